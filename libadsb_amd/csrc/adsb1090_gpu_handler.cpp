// adsb1090_gpu_handler.cpp -- libadsb's ADSB1090Handler surface on top of the C ABI (include/adsb_amd.h).
//
// Drop-in for the translation unit ADSB1090.cpp of libadsb: it defines the same two factories
// (reference ADSB.h:13-15, 24-26; definitions ADSB1090.cpp:1253-1265) and an object that implements
// RTLSDR::IDataHandler + ADSB::IDataProvider exactly as the reference handler does (ADSB1090.cpp:99, 158-190):
//   HandleData(span)          synchronous; callbacks fire inside the call, in sample order   (:158-175, 1173)
//   OnDeviceStatusChanged(b)  forwarded to the listener with the source id                  (:177)
//   Start / Stop              keep the listener; inside libadsb they also start/stop the RTLSDR transport (:179-188)
//   NotifySelfLocation        ignored, as in the reference                                    (:190)
// Errors follow the reference's conventions: construction throws (std::runtime_error) when no GPU context can be
// created -- there is no CPU fallback --, HandleData never throws across the transport's noexcept trampoline
// (RTLSDR.hpp:549-555): a failing scan is reported on stderr and the buffer is dropped.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

#include "adsb_amd.h"
#include "libadsb_iface.hpp"

namespace
{
struct ADSB1090GpuHandler final : RTLSDR::IDataHandler, ADSB::IDataProvider
{
    ADSB1090GpuHandler(std::shared_ptr<ADSB::TrafficManager> tm, RTLSDR::IDeviceSelector const* /*selector*/, ADSB::Source source)
        : trafficManager(std::move(tm)), sourceId(source)
    {
        if (int rc = adsb_amd_handler_create(&gpu, -1); rc != ADSB_AMD_OK)
            throw std::runtime_error(std::string("libadsb_amd: ") + adsb_amd_handler_last_error(nullptr));
#ifdef LIBADSB_AMD_WITH_LIBADSB_HEADERS
        // in-tree: transport = std::make_unique<RTLSDR>(selector, RTLSDR::Config{.frequency = 1090000000, .sampleRate = 2000000});
#endif
    }
    ~ADSB1090GpuHandler() override { adsb_amd_handler_destroy(gpu); }
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
    void Start(ADSB::IListener& l) override { listener = &l; }
    void Stop() override {}
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
