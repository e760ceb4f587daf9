// test_1090_gpu.cpp -- the reference test's flow (tests/test_1090.cpp:55-72, 103-123) against the GPU handler:
// create the handler through ADSB::test::TryCreateADSB1090Handler, push buffers through HandleData, print one line
// per OnChanged callback in the reference's format.  The Python test compares the lines with the oracle's.
//   usage: test_1090_gpu <iq file> <buffer bytes (0 = whole file in one call)>
//          test_1090_gpu --provider <callbacks to wait for>     the production factory (ADSB::TryCreateADSB1090Handler, ADSB.h:13-15)
//              started like DataProviderImpl does (ADSBListener.cpp:25-29): with "1090000000.test.dat" in the working directory the
//              handler's transport replays it (RTLSDR.hpp:396-442) and the callbacks arrive on its consumer thread; the driver prints
//              the first <n> and stops the provider
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "libadsb_iface.hpp"

static std::atomic<long> g_callbacks{0};
static long              g_limit = -1; // --provider: stop printing after this many (the replay loops, as the reference's does)

struct Listener : ADSB::IListener
{
    void OnChanged(ADSB::IAirCraft const& a) override
    {
        if (g_limit >= 0 && g_callbacks.load() >= g_limit) return;
        g_callbacks.fetch_add(1);
        // "{:x}[{: >8}]: Pos={:+03.2f}:{:+03.2f}^{:05} Speed={:03} Count={}" with Count bound to the squawk
        auto cs = a.FlightNumber();
        std::printf("%x[", a.Addr());
        std::fwrite(cs.data(), 1, cs.size(), stdout);
        std::printf("]: Pos=%+03.2f:%+03.2f^%05d Speed=%03u Count=%u\n", a.Lat1E7() / 10000000., a.Lon1E7() / 10000000., a.Altitude(), a.Speed(),
                    a.SquakCode());
    }
    void OnDeviceStatusChanged(ADSB::Source, bool) override {}
};
struct Selector : RTLSDR::IDeviceSelector
{
    [[nodiscard]] bool SelectDevice(RTLSDR::DeviceInfo const&) const override { return false; }
};

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    if (std::strcmp(argv[1], "--provider") == 0)
    {
        g_limit = std::strtol(argv[2], nullptr, 10);
        Listener listener;
        auto     mgr = std::make_shared<ADSB::TrafficManager>();
        mgr->SetListener(&listener);
        Selector selector;
        auto     provider = ADSB::TryCreateADSB1090Handler(mgr, &selector, ADSB::Source::ADSB1090);
        provider->Start(listener);
        for (int waited = 0; g_callbacks.load() < g_limit && waited < 60000; waited += 5) std::this_thread::sleep_for(std::chrono::milliseconds(5));
        provider->Stop();
        provider->Start(listener); // a second Start after Stop must work (RTLSDR::Start joins the old threads, :444-455) ...
        provider->Stop();          // ... and so must stopping again
        std::fflush(stdout);
        return g_callbacks.load() >= g_limit ? 0 : 3;
    }
    std::ifstream        f(argv[1], std::ios::binary);
    std::vector<uint8_t> iq((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    size_t               bb = std::strtoull(argv[2], nullptr, 10);
    Listener             listener;
    auto                 mgr = std::make_shared<ADSB::TrafficManager>();
    mgr->SetListener(&listener);
    Selector selector;
    auto     handler = ADSB::test::TryCreateADSB1090Handler(mgr, &selector, ADSB::Source::ADSB1090);
    if (bb == 0) handler->HandleData({iq.data(), iq.size()});
    else
        for (size_t o = 0; o + bb <= iq.size(); o += bb) handler->HandleData({iq.data() + o, bb}); // trailing partial chunk dropped (:87-95)
    return 0;
}
