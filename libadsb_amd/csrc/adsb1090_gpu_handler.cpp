// adsb1090_gpu_handler.cpp -- libadsb's ADSB1090Handler surface on top of the C ABI (include/adsb_amd.h).
//
// Drop-in for the translation unit ADSB1090.cpp of libadsb: it defines the same two factories
// (reference ADSB.h:13-15, 24-26; definitions ADSB1090.cpp:1253-1265) and an object that implements
// RTLSDR::IDataHandler + ADSB::IDataProvider exactly as the reference handler does (ADSB1090.cpp:99, 158-190):
//   HandleData(span)          synchronous; callbacks fire inside the call, in sample order   (:158-175, 1173)
//   OnDeviceStatusChanged(b)  forwarded to the listener with the source id                  (:177)
//   Start / Stop              keep the listener and start / stop the transport (:179-188)
//   NotifySelfLocation        ignored, as in the reference                                    (:190)
// The transport: inside a libadsb checkout (LIBADSB_AMD_WITH_LIBADSB_HEADERS) the handler owns an RTLSDR object built with the
// reference's Config{.frequency = 1090000000, .sampleRate = 2000000} and the caller's selector (:148) and forwards Start/Stop to
// it (:179-188) -- that branch needs librtlsdr's header and cannot be compiled in this repository's image.  Stand-alone the
// handler owns adsb_amd::Transport (transport.hpp), which gives it what RTLSDR gives a handler without a receiver attached: the
// replay of "1090000000.test.dat" from the working directory through the 16 x 262144-byte ring (RTLSDR.hpp:396-442) and, for a host
// that owns the receiver, a push entry point.  With neither a recording nor a producer the consumer thread simply waits, as
// the reference's does until a device shows up.
// Errors follow the reference's conventions: construction throws (std::runtime_error) when no GPU context can be
// created -- there is no CPU fallback --, HandleData never throws across the transport's noexcept trampoline
// (RTLSDR.hpp:549-555): a failing scan is reported on stderr and the buffer is dropped.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

#include "adsb_amd.h"
#include "libadsb_iface.hpp"
#ifndef LIBADSB_AMD_WITH_LIBADSB_HEADERS
#include "transport.hpp"
#endif

namespace
{
#ifdef LIBADSB_AMD_WITH_LIBADSB_HEADERS
struct ADSB1090GpuHandler final : RTLSDR::IDataHandler, ADSB::IDataProvider
{
    ADSB1090GpuHandler(std::shared_ptr<ADSB::TrafficManager> tm, RTLSDR::IDeviceSelector const* selector, ADSB::Source source)
        : trafficManager(std::move(tm)), sourceId(source), listener1090{selector, RTLSDR::Config{.frequency = 1090000000, .sampleRate = 2000000}}
    {
        if (int rc = adsb_amd_handler_create(&gpu, -1); rc != ADSB_AMD_OK)
            throw std::runtime_error(std::string("libadsb_amd: ") + adsb_amd_handler_last_error(nullptr));
    }
    void StartTransport() { listener1090.Start(this); } // ADSB1090.cpp:182
    void StopTransport() { listener1090.Stop(); }       // :187
#else
struct ADSB1090GpuHandler final : RTLSDR::IDataHandler, ADSB::IDataProvider, adsb_amd::Transport::Sink
{
    ADSB1090GpuHandler(std::shared_ptr<ADSB::TrafficManager> tm, RTLSDR::IDeviceSelector const* /*selector: no receiver stand-alone*/,
                       ADSB::Source source)
        : trafficManager(std::move(tm)), sourceId(source), transport(adsb_amd::Transport::ReplayFileFor(1090000000))
    {
        if (int rc = adsb_amd_handler_create(&gpu, -1); rc != ADSB_AMD_OK)
            throw std::runtime_error(std::string("libadsb_amd: ") + adsb_amd_handler_last_error(nullptr));
    }
    void StartTransport() { transport.Start(this); }
    void StopTransport() { transport.Stop(); }
    void Deliver(const uint8_t* data, size_t nbytes) override { HandleData(std::span<uint8_t const>(data, nbytes)); } // RTLSDR.hpp:531
#endif
    ~ADSB1090GpuHandler() override
    {
        StopTransport(); // no HandleData may be running when the GPU context goes away
        adsb_amd_handler_destroy(gpu);
    }
    ADSB1090GpuHandler(ADSB1090GpuHandler const&)            = delete;
    ADSB1090GpuHandler& operator=(ADSB1090GpuHandler const&) = delete;

    void HandleData(std::span<uint8_t const> const& data) override
    {
        // one call = one independent buffer, whatever its size (reference behaviour; live buffers are 262144 bytes)
        long n = adsb_amd_handler_handle_data(gpu, data.data(), data.size(), 0, &ADSB1090GpuHandler::OnFrame, this);
        if (n < 0) std::fprintf(stderr, "libadsb_amd: HandleData dropped a buffer: %s\n", adsb_amd_handler_last_error(gpu));
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

    // one accepted frame: mirror the snapshot into the traffic manager's record and notify (ADSB1090.cpp:1129-1173)
    static void OnFrame(void* user, const adsb_amd_frame_t* /*frame*/, const adsb_amd_aircraft_t* s)
    {
        auto* self = static_cast<ADSB1090GpuHandler*>(user);
        auto& a    = self->trafficManager->FindOrCreate(s->addr);
        a.sourceId = self->sourceId;
        std::memcpy(a.callsign.data(), s->callsign, 8);
        a.altitude = s->altitude;
        a.speed    = s->speed;
        a.track    = s->track;
        a.lat1E7   = s->lat1e7;
        a.lon1E7   = s->lon1e7;
        self->trafficManager->NotifyChanged(a);
    }

    adsb_amd_handler_t*                   gpu = nullptr;
    std::shared_ptr<ADSB::TrafficManager> trafficManager;
    ADSB::IListener*                      listener = nullptr;
    ADSB::Source                          sourceId;
#ifdef LIBADSB_AMD_WITH_LIBADSB_HEADERS
    RTLSDR listener1090; // declared last, as in the reference (:232): constructed after, destroyed before everything it calls into
#else
    adsb_amd::Transport transport;
#endif
};
} // namespace

std::unique_ptr<ADSB::IDataProvider> ADSB::TryCreateADSB1090Handler(std::shared_ptr<ADSB::TrafficManager> const& trafficManager,
                                                                    RTLSDR::IDeviceSelector const* selector, ADSB::Source sourceId)
{
    return std::make_unique<ADSB1090GpuHandler>(trafficManager, selector, sourceId);
}

std::unique_ptr<RTLSDR::IDataHandler> ADSB::test::TryCreateADSB1090Handler(std::shared_ptr<ADSB::TrafficManager> const& trafficManager,
                                                                           RTLSDR::IDeviceSelector const* selector, ADSB::Source sourceId)
{
    return std::make_unique<ADSB1090GpuHandler>(trafficManager, selector, sourceId);
}
