// libadsb_iface.hpp -- the slice of libadsb's public surface that the GPU 1090 handler plugs into.
//
// Inside a libadsb checkout define LIBADSB_AMD_WITH_LIBADSB_HEADERS and this header simply pulls in libadsb's own
// ADSB.h (ADSBListener.h, AircraftImpl.h, RTLSDR.hpp).  Stand-alone (this repository's tests, no librtlsdr) it declares
// the same names with the same signatures, restated from the interface contract -- what a listener can observe:
//   ADSB::IAirCraft / IListener / IDataProvider      (reference ADSBListener.h:27-72)
//   ADSB::AirCraftImpl / TrafficManager               (reference AircraftImpl.h:9-68)
//   RTLSDR::IDataHandler / IDeviceSelector            (reference RTLSDR.hpp:39-49, 58-75)
//   ADSB::TryCreateADSB1090Handler, ADSB::test::...   (reference ADSB.h:13-15, 24-26)
#pragma once

#ifdef LIBADSB_AMD_WITH_LIBADSB_HEADERS
#include "ADSB.h"
#else

#include <array>
#include <chrono>
#include <cstdint>
#include <memory>
#include <span>
#include <string_view>
#include <unordered_map>

namespace ADSB
{
enum class Source : uint8_t
{
    UAT978        = 1u,
    ADSB1090      = 2u,
    FlightRadar24 = 4u,
};

struct IAirCraft
{
    using time_point = std::chrono::time_point<std::chrono::system_clock>;
    virtual ~IAirCraft() = default;
    [[nodiscard]] virtual Source           SourceId() const     = 0;
    [[nodiscard]] virtual uint32_t         MessageCount() const = 0;
    [[nodiscard]] virtual uint32_t         Addr() const         = 0;
    [[nodiscard]] virtual std::string_view FlightNumber() const = 0;
    [[nodiscard]] virtual time_point       LastSeen() const     = 0;
    [[nodiscard]] virtual uint32_t         SquakCode() const    = 0;
    [[nodiscard]] virtual int32_t          Altitude() const     = 0;
    [[nodiscard]] virtual uint32_t         Speed() const        = 0;
    [[nodiscard]] virtual uint32_t         Heading() const      = 0;
    [[nodiscard]] virtual int32_t          Climb() const        = 0;
    [[nodiscard]] virtual int32_t          Lat1E7() const       = 0;
    [[nodiscard]] virtual int32_t          Lon1E7() const       = 0;
};

struct IListener
{
    virtual ~IListener()                                                = default;
    virtual void OnChanged(IAirCraft const&)                            = 0;
    virtual void OnDeviceStatusChanged(Source sourceId, bool available) = 0;
};

struct IDataProvider
{
    virtual ~IDataProvider()                          = default;
    virtual void Start(IListener& listener)           = 0;
    virtual void Stop()                               = 0;
    virtual void NotifySelfLocation(IAirCraft const&) = 0;
};

// State record the traffic manager owns: the reference's members in the reference's order (AircraftImpl.h:26-44), so that the object has the same
// size and every member the same offset under either header (held equal at compile time by tests/cpp/iface_matches_reference.cpp).  The six cpr*
// members are the reference demodulator's scratch for its even/odd pair; the GPU handler keeps that state in its resolver and leaves them untouched.
struct AirCraftImpl : IAirCraft
{
    [[nodiscard]] Source           SourceId() const override { return sourceId; }
    [[nodiscard]] uint32_t         MessageCount() const override { return 0; }
    [[nodiscard]] uint32_t         Addr() const override { return addr; }
    [[nodiscard]] std::string_view FlightNumber() const override { return {callsign.data(), callsign.size()}; }
    [[nodiscard]] time_point       LastSeen() const override { return seen; }
    [[nodiscard]] uint32_t         SquakCode() const override { return modeA; }
    [[nodiscard]] int32_t          Altitude() const override { return altitude; }
    [[nodiscard]] uint32_t         Speed() const override { return speed; }
    [[nodiscard]] uint32_t         Heading() const override { return track; }
    [[nodiscard]] int32_t          Climb() const override { return vertRate; }
    [[nodiscard]] int32_t          Lat1E7() const override { return lat1E7; }
    [[nodiscard]] int32_t          Lon1E7() const override { return lon1E7; }

    uint32_t            addr{0};
    std::array<char, 8> callsign{};
    time_point          seen{};
    uint32_t            modeA{};
    int32_t             altitude{};
    uint32_t            speed{};
    uint32_t            track{};
    int32_t             vertRate{};
    int32_t             lat1E7{};
    int32_t             lon1E7{};
    double              cprOddLat{};
    double              cprOddLon{};
    time_point          cprOddTime{};
    double              cprEvenLat{};
    double              cprEvenLon{};
    time_point          cprEvenTime{};
    Source              sourceId{};
};

struct TrafficManager : std::enable_shared_from_this<TrafficManager>
{
    AirCraftImpl& FindOrCreate(uint32_t addr)
    {
        auto& slot = aircrafts[addr];
        if (!slot)
        {
            slot       = std::make_unique<AirCraftImpl>();
            slot->addr = addr;
        }
        return *slot;
    }
    void SetListener(IListener* l) { listener = l; }
    void NotifyChanged(AirCraftImpl const& a) const { listener->OnChanged(a); }

    std::unordered_map<uint32_t, std::unique_ptr<AirCraftImpl>> aircrafts;
    IListener*                                                  listener{nullptr};
};
} // namespace ADSB

struct RTLSDR
{
    static constexpr size_t BufferLength = size_t{65536u} * 4u;
    static constexpr size_t BufferCount  = 16;
    struct DeviceInfo
    { // same members in the same order as the reference's (RTLSDR.hpp:67-73): a selector compiled against either header reads the same bytes
        uint32_t index;
        char     vendor[256];
        char     product[256];
        char     serial[256];
    };
    struct IDeviceSelector
    {
        virtual ~IDeviceSelector()                                       = default;
        [[nodiscard]] virtual bool SelectDevice(DeviceInfo const&) const = 0;
    };
    struct IDataHandler
    {
        virtual ~IDataHandler()                                       = default;
        virtual void HandleData(std::span<uint8_t const> const& data) = 0;
        virtual void OnDeviceStatusChanged(bool available)            = 0;
    };
};

namespace ADSB
{
std::unique_ptr<IDataProvider> TryCreateADSB1090Handler(std::shared_ptr<TrafficManager> const& trafficManager,
                                                        RTLSDR::IDeviceSelector const* selector, Source sourceId);
// reference ADSB.h:10-12, 16, 21-23 (definitions UAT978.cpp:114-126, 12-19)
std::unique_ptr<IDataProvider> TryCreateUAT978Handler(std::shared_ptr<TrafficManager> const& trafficManager,
                                                      RTLSDR::IDeviceSelector const* selector, Source sourceId);
TrafficManager**               GetThreadLocalTrafficManager();
namespace test
{
std::unique_ptr<RTLSDR::IDataHandler> TryCreateADSB1090Handler(std::shared_ptr<TrafficManager> const& trafficManager,
                                                               RTLSDR::IDeviceSelector const* selector, Source sourceId);
std::unique_ptr<RTLSDR::IDataHandler> TryCreateUAT978Handler(std::shared_ptr<TrafficManager> const& trafficManager,
                                                             RTLSDR::IDeviceSelector const* selector, Source sourceId);
}
} // namespace ADSB
#endif
