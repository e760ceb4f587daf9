// uat978_gpu_handler.cpp -- libadsb's UAT978Handler surface on top of the C ABI (include/adsb_amd.h, "UAT 978").
//
// Drop-in for the translation unit UAT978.cpp of libadsb *and* for the dump978 legacy objects it links
// (CMakeLists.txt:87-95): it defines the same two factories (reference ADSB.h, definitions UAT978.cpp:114-126), the
// thread-local traffic-manager slot the up-call reads (UAT978.cpp:12-19), and an object that implements
// RTLSDR::IDataHandler + ADSB::IDataProvider as the reference handler does (UAT978.cpp:21-111):
//   HandleData(span)          publishes the traffic manager in the thread-local slot (:46), runs the staging rounds of :48-59
//                             on the GPU (including :57's half-tail carry) and calls the host's
//                             dump_raw_message(updown, data, len, rs_errors) (uat2json-wrapper.cpp:14) once per frame, in
//                             stream order, before returning
//   OnDeviceStatusChanged(b)  forwarded to the listener with the source id (:62)
//   Start / Stop              keep the listener and start / stop the transport (:64-72): inside a libadsb checkout
//                             (LIBADSB_AMD_WITH_LIBADSB_HEADERS; not compilable in this repository's image, it needs librtlsdr's
//                             header) an RTLSDR member built with Config{.gain = 48, .frequency = 978000000, .sampleRate = 2083334}
//                             (:29); stand-alone adsb_amd::Transport (transport.hpp): the replay of "978000000.test.dat" from the
//                             working directory through the 16 x 262144-byte ring (RTLSDR.hpp:396-442)
//   NotifySelfLocation        ignored, as in the reference (:74)
// dump_raw_message and uat_decode_adsb_mdb stay the host's (uat2json-wrapper.cpp and dump978's uat_decode.c are not
// part of the demodulation path).  Construction throws when no GPU context can be created: there is no CPU fallback.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

#include "adsb_amd.h"
#include "libadsb_iface.hpp"
#ifndef LIBADSB_AMD_WITH_LIBADSB_HEADERS
#include "transport.hpp"
#endif

extern "C" void dump_raw_message(char updown, uint8_t* data, int len, int rs_errors) __attribute__((weak));

ADSB::TrafficManager** ADSB::GetThreadLocalTrafficManager()
{
    static thread_local TrafficManager* slot;
    return &slot;
}

namespace
{
#ifdef LIBADSB_AMD_WITH_LIBADSB_HEADERS
struct UAT978GpuHandler final : RTLSDR::IDataHandler, ADSB::IDataProvider
{
    UAT978GpuHandler(std::shared_ptr<ADSB::TrafficManager> tm, RTLSDR::IDeviceSelector const* selector, ADSB::Source source)
        : trafficManager(std::move(tm)), sourceId(source),
          listener978{selector, RTLSDR::Config{.gain = 48, .frequency = 978000000, .sampleRate = 2083334}}
    {
        if (int rc = adsb_amd_uat_create(&gpu, 0); rc != ADSB_AMD_OK)
            throw std::runtime_error(std::string("libadsb_amd: ") + adsb_amd_uat_last_error(nullptr));
    }
    void StartTransport() { listener978.Start(this); } // UAT978.cpp:67
    void StopTransport() { listener978.Stop(); }       // :71
#else
struct UAT978GpuHandler final : RTLSDR::IDataHandler, ADSB::IDataProvider, adsb_amd::Transport::Sink
{
    UAT978GpuHandler(std::shared_ptr<ADSB::TrafficManager> tm, RTLSDR::IDeviceSelector const* /*selector: no receiver stand-alone*/,
                     ADSB::Source source)
        : trafficManager(std::move(tm)), sourceId(source), transport(adsb_amd::Transport::ReplayFileFor(978000000))
    {
        if (int rc = adsb_amd_uat_create(&gpu, 0); rc != ADSB_AMD_OK)
            throw std::runtime_error(std::string("libadsb_amd: ") + adsb_amd_uat_last_error(nullptr));
    }
    void StartTransport() { transport.Start(this); }
    void StopTransport() { transport.Stop(); }
    void Deliver(const uint8_t* data, size_t nbytes) override { HandleData(std::span<uint8_t const>(data, nbytes)); } // RTLSDR.hpp:531
#endif
    ~UAT978GpuHandler() override
    {
        StopTransport();
        adsb_amd_uat_destroy(gpu);
    }
    UAT978GpuHandler(UAT978GpuHandler const&)            = delete;
    UAT978GpuHandler& operator=(UAT978GpuHandler const&) = delete;

    void HandleData(std::span<uint8_t const> const& data) override
    {
        *ADSB::GetThreadLocalTrafficManager() = trafficManager.get();
        if (adsb_amd_uat_handle_data(gpu, data.data(), data.size(), &UAT978GpuHandler::OnFrame, this) != ADSB_AMD_OK)
            std::fprintf(stderr, "libadsb_amd: UAT HandleData dropped a buffer: %s\n", adsb_amd_uat_last_error(gpu));
    }
    void OnDeviceStatusChanged(bool available) override
    {
        if (listener) listener->OnDeviceStatusChanged(sourceId, available);
    }
    void Start(ADSB::IListener& l) override
    {
        listener = &l;
        StartTransport();
    }
    void Stop() override { StopTransport(); }
    void NotifySelfLocation(ADSB::IAirCraft const& /*unused*/) override {}

    static void OnFrame(void* /*user*/, char updown, const uint8_t* data, int len, int rs_errors, uint64_t /*sample_index*/)
    {
        if (!dump_raw_message) return; // a host without the up-call (nothing to deliver to)
        uint8_t copy[432];
        std::memcpy(copy, data, (size_t)len);
        dump_raw_message(updown, copy, len, rs_errors);
    }

    std::shared_ptr<ADSB::TrafficManager> trafficManager;
    ADSB::IListener*                      listener = nullptr;
    ADSB::Source                          sourceId;
    adsb_amd_uat_t*                       gpu = nullptr;
#ifdef LIBADSB_AMD_WITH_LIBADSB_HEADERS
    RTLSDR listener978;
#else
    adsb_amd::Transport transport;
#endif
};
} // namespace

std::unique_ptr<ADSB::IDataProvider> ADSB::TryCreateUAT978Handler(std::shared_ptr<ADSB::TrafficManager> const& trafficManager,
                                                                  RTLSDR::IDeviceSelector const* selector, ADSB::Source sourceId)
{
    return std::make_unique<UAT978GpuHandler>(trafficManager, selector, sourceId);
}

std::unique_ptr<RTLSDR::IDataHandler> ADSB::test::TryCreateUAT978Handler(std::shared_ptr<ADSB::TrafficManager> const& trafficManager,
                                                                         RTLSDR::IDeviceSelector const* selector, ADSB::Source sourceId)
{
    return std::make_unique<UAT978GpuHandler>(trafficManager, selector, sourceId);
}
