// iface_matches_reference.cpp -- build-container check (tests/test_iface_matches_reference.py): the stand-alone slice of libadsb's surface in
// include/libadsb_iface.hpp against the reference's own ADSBListener.h and AircraftImpl.h, in one translation unit.  The two reference headers need
// only CommonMacros.h (no <rtl-sdr.h>, no patch); they are included inside `namespace ref` -- after every standard header they pull in, so that
// those stay at global scope -- and every name, signature, enumerator, member offset and size the two sides share is held equal at compile time;
// main() then calls every virtual function of the stand-alone interfaces THROUGH the reference's types (same object, reinterpreted), which fails
// if the order of the virtual functions ever differs.  ADSB.h / RTLSDR.hpp cannot be compiled here (they include <rtl-sdr.h>, which this image lacks).
//
// Compile: g++ -std=c++20 -Wno-invalid-offsetof -I include -I /root/reference tests/cpp/iface_matches_reference.cpp
#include <array>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <span>
#include <string_view>
#include <type_traits>
#include <unordered_map>

#include "libadsb_iface.hpp" // stand-alone branch (LIBADSB_AMD_WITH_LIBADSB_HEADERS not defined)

namespace ref
{
#include "ADSBListener.h" // /root/reference/ADSBListener.h:27-72
#include "AircraftImpl.h" // /root/reference/AircraftImpl.h:9-68
} // namespace ref

namespace mine = ::ADSB;
namespace theirs = ::ref::ADSB;

// ---- a reference type and the stand-alone type that stands for it
template <class T> struct mirror { using type = T; };
template <> struct mirror<theirs::Source> { using type = mine::Source; };
template <> struct mirror<theirs::IAirCraft> { using type = mine::IAirCraft; };
template <> struct mirror<theirs::IListener> { using type = mine::IListener; };
template <> struct mirror<theirs::IDataProvider> { using type = mine::IDataProvider; };
template <> struct mirror<theirs::AirCraftImpl> { using type = mine::AirCraftImpl; };
template <> struct mirror<theirs::TrafficManager> { using type = mine::TrafficManager; };
template <class T> struct mirror<T const> { using type = typename mirror<T>::type const; };
template <class T> struct mirror<T&> { using type = typename mirror<T>::type&; };
template <class T> struct mirror<T*> { using type = typename mirror<T>::type*; };
template <class T> struct mirror<std::unique_ptr<T>> { using type = std::unique_ptr<typename mirror<T>::type>; };
template <class K, class V> struct mirror<std::unordered_map<K, V>> { using type = std::unordered_map<K, typename mirror<V>::type>; };
template <class T> using mirror_t = typename mirror<T>::type;

// ---- member-function pointer -> its signature with the class taken out (and the reference's types mirrored)
template <class F> struct signature;
template <class C, class R, class... A> struct signature<R (C::*)(A...)> { using type = mirror_t<R>(mirror_t<A>...); static constexpr bool is_const = false; };
template <class C, class R, class... A> struct signature<R (C::*)(A...) const> { using type = mirror_t<R>(mirror_t<A>...); static constexpr bool is_const = true; };
template <class F, class G>
constexpr bool same_member = std::is_same_v<typename signature<F>::type, typename signature<G>::type> && signature<F>::is_const == signature<G>::is_const;
#define SAME_MEMBER(Class, name) static_assert(same_member<decltype(&theirs::Class::name), decltype(&mine::Class::name)>, #Class "::" #name " differs from the reference's")
#define SAME_FIELD(Class, name)                                                                                                                    \
    static_assert(std::is_same_v<mirror_t<decltype(theirs::Class::name)>, decltype(mine::Class::name)>, #Class "::" #name ": type differs from the reference's"); \
    static_assert(offsetof(theirs::Class, name) == offsetof(mine::Class, name), #Class "::" #name ": offset differs from the reference's")

// Source (ADSBListener.h:14-19)
static_assert(std::is_same_v<std::underlying_type_t<theirs::Source>, std::underlying_type_t<mine::Source>>);
static_assert((uint8_t)theirs::Source::UAT978 == (uint8_t)mine::Source::UAT978 && (uint8_t)theirs::Source::ADSB1090 == (uint8_t)mine::Source::ADSB1090 &&
              (uint8_t)theirs::Source::FlightRadar24 == (uint8_t)mine::Source::FlightRadar24);

// IAirCraft (ADSBListener.h:27-50)
static_assert(std::is_same_v<theirs::IAirCraft::time_point, mine::IAirCraft::time_point>);
SAME_MEMBER(IAirCraft, SourceId);
SAME_MEMBER(IAirCraft, MessageCount);
SAME_MEMBER(IAirCraft, Addr);
SAME_MEMBER(IAirCraft, FlightNumber);
SAME_MEMBER(IAirCraft, LastSeen);
SAME_MEMBER(IAirCraft, SquakCode);
SAME_MEMBER(IAirCraft, Altitude);
SAME_MEMBER(IAirCraft, Speed);
SAME_MEMBER(IAirCraft, Heading);
SAME_MEMBER(IAirCraft, Climb);
SAME_MEMBER(IAirCraft, Lat1E7);
SAME_MEMBER(IAirCraft, Lon1E7);
static_assert(sizeof(theirs::IAirCraft) == sizeof(mine::IAirCraft) && std::is_abstract_v<mine::IAirCraft> && std::has_virtual_destructor_v<mine::IAirCraft>);
// IListener (:52-60), IDataProvider (:62-72)
SAME_MEMBER(IListener, OnChanged);
SAME_MEMBER(IListener, OnDeviceStatusChanged);
SAME_MEMBER(IDataProvider, Start);
SAME_MEMBER(IDataProvider, Stop);
SAME_MEMBER(IDataProvider, NotifySelfLocation);
static_assert(sizeof(theirs::IListener) == sizeof(mine::IListener) && sizeof(theirs::IDataProvider) == sizeof(mine::IDataProvider));
static_assert(std::has_virtual_destructor_v<mine::IListener> && std::has_virtual_destructor_v<mine::IDataProvider>);

// AirCraftImpl (AircraftImpl.h:9-44): every data member at the reference's offset, the object the reference's size
static_assert(std::is_base_of_v<mine::IAirCraft, mine::AirCraftImpl> && sizeof(theirs::AirCraftImpl) == sizeof(mine::AirCraftImpl) &&
              alignof(theirs::AirCraftImpl) == alignof(mine::AirCraftImpl));
SAME_FIELD(AirCraftImpl, addr);
SAME_FIELD(AirCraftImpl, callsign);
SAME_FIELD(AirCraftImpl, seen);
SAME_FIELD(AirCraftImpl, modeA);
SAME_FIELD(AirCraftImpl, altitude);
SAME_FIELD(AirCraftImpl, speed);
SAME_FIELD(AirCraftImpl, track);
SAME_FIELD(AirCraftImpl, vertRate);
SAME_FIELD(AirCraftImpl, lat1E7);
SAME_FIELD(AirCraftImpl, lon1E7);
SAME_FIELD(AirCraftImpl, cprOddLat);
SAME_FIELD(AirCraftImpl, cprOddLon);
SAME_FIELD(AirCraftImpl, cprOddTime);
SAME_FIELD(AirCraftImpl, cprEvenLat);
SAME_FIELD(AirCraftImpl, cprEvenLon);
SAME_FIELD(AirCraftImpl, cprEvenTime);
SAME_FIELD(AirCraftImpl, sourceId);

// TrafficManager (AircraftImpl.h:46-67)
static_assert(std::is_base_of_v<std::enable_shared_from_this<mine::TrafficManager>, mine::TrafficManager>);
SAME_MEMBER(TrafficManager, FindOrCreate);
SAME_MEMBER(TrafficManager, SetListener);
SAME_MEMBER(TrafficManager, NotifyChanged);
SAME_FIELD(TrafficManager, aircrafts);
SAME_FIELD(TrafficManager, listener);
static_assert(sizeof(theirs::TrafficManager) == sizeof(mine::TrafficManager));

// ---- the order of the virtual functions: objects of the stand-alone types, called through the reference's
namespace
{
struct ProbeAircraft : mine::IAirCraft
{
    mine::Source     SourceId() const override { return mine::Source::FlightRadar24; }
    uint32_t         MessageCount() const override { return 101; }
    uint32_t         Addr() const override { return 102; }
    std::string_view FlightNumber() const override { return "PROBE103"; }
    time_point       LastSeen() const override { return time_point{std::chrono::seconds{104}}; }
    uint32_t         SquakCode() const override { return 105; }
    int32_t          Altitude() const override { return 106; }
    uint32_t         Speed() const override { return 107; }
    uint32_t         Heading() const override { return 108; }
    int32_t          Climb() const override { return 109; }
    int32_t          Lat1E7() const override { return 110; }
    int32_t          Lon1E7() const override { return 111; }
};
struct ProbeListener : mine::IListener
{
    int  changed = 0, status = 0;
    void OnChanged(mine::IAirCraft const& a) override { changed += (int)a.Addr(); }
    void OnDeviceStatusChanged(mine::Source s, bool available) override { status += (int)s * 10 + (available ? 1 : 0); }
};
struct ProbeProvider : mine::IDataProvider
{
    int  started = 0, stopped = 0, located = 0;
    void Start(mine::IListener&) override { started++; }
    void Stop() override { stopped++; }
    void NotifySelfLocation(mine::IAirCraft const& a) override { located += (int)a.Altitude(); }
};
int failures = 0;
void expect(bool ok, const char* what)
{
    if (!ok) std::printf("MISMATCH: %s\n", what), failures++;
}
} // namespace

int main()
{
    ProbeAircraft pa;
    auto const&   ra = *reinterpret_cast<theirs::IAirCraft const*>(static_cast<mine::IAirCraft const*>(&pa));
    expect(ra.SourceId() == theirs::Source::FlightRadar24, "IAirCraft::SourceId slot");
    expect(ra.MessageCount() == 101, "IAirCraft::MessageCount slot");
    expect(ra.Addr() == 102, "IAirCraft::Addr slot");
    expect(ra.FlightNumber() == "PROBE103", "IAirCraft::FlightNumber slot");
    expect(ra.LastSeen() == theirs::IAirCraft::time_point{std::chrono::seconds{104}}, "IAirCraft::LastSeen slot");
    expect(ra.SquakCode() == 105, "IAirCraft::SquakCode slot");
    expect(ra.Altitude() == 106, "IAirCraft::Altitude slot");
    expect(ra.Speed() == 107, "IAirCraft::Speed slot");
    expect(ra.Heading() == 108, "IAirCraft::Heading slot");
    expect(ra.Climb() == 109, "IAirCraft::Climb slot");
    expect(ra.Lat1E7() == 110, "IAirCraft::Lat1E7 slot");
    expect(ra.Lon1E7() == 111, "IAirCraft::Lon1E7 slot");

    ProbeListener pl;
    auto&         rl = *reinterpret_cast<theirs::IListener*>(static_cast<mine::IListener*>(&pl));
    rl.OnChanged(ra);
    rl.OnDeviceStatusChanged(theirs::Source::ADSB1090, true);
    expect(pl.changed == 102 && pl.status == 21, "IListener slots");

    ProbeProvider pp;
    auto&         rp = *reinterpret_cast<theirs::IDataProvider*>(static_cast<mine::IDataProvider*>(&pp));
    rp.Start(rl);
    rp.NotifySelfLocation(ra);
    rp.Stop();
    expect(pp.started == 1 && pp.stopped == 1 && pp.located == 106, "IDataProvider slots");

    // a record of the stand-alone traffic manager read through the reference's AirCraftImpl, and its listener hook through the reference's manager
    auto  tm = std::make_shared<mine::TrafficManager>();
    auto& a  = tm->FindOrCreate(0x484412);
    a.altitude = 38000, a.lat1E7 = 522657801, a.lon1E7 = 39389125, a.sourceId = mine::Source::ADSB1090, a.cprOddLat = 1.5;
    auto const& r = *reinterpret_cast<theirs::AirCraftImpl const*>(&a);
    expect(r.Addr() == 0x484412 && r.Altitude() == 38000 && r.Lat1E7() == 522657801 && r.Lon1E7() == 39389125 && r.SourceId() == theirs::Source::ADSB1090 &&
               r.cprOddLat == 1.5,
           "AirCraftImpl read through the reference's type");
    auto& rtm = *reinterpret_cast<theirs::TrafficManager*>(tm.get());
    rtm.SetListener(&rl);
    rtm.NotifyChanged(r);
    expect(pl.changed == 102 + 0x484412, "TrafficManager::NotifyChanged through the reference's type");
    auto& again = rtm.FindOrCreate(0x484412);
    expect(reinterpret_cast<void*>(&again) == reinterpret_cast<void*>(&a), "TrafficManager::FindOrCreate through the reference's type finds the same record");
    if (failures == 0) std::printf("iface ok: libadsb_iface.hpp == reference ADSBListener.h + AircraftImpl.h (signatures, layout, virtual order)\n");
    return failures ? 1 : 0;
}
