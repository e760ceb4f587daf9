// test_1090_gpu.cpp -- the reference test's flow (tests/test_1090.cpp:55-72, 103-123) against the GPU handler:
// create the handler through ADSB::test::TryCreateADSB1090Handler, push buffers through HandleData, print one line
// per OnChanged callback in the reference's format.  The Python test compares the lines with the oracle's.
//   usage: test_1090_gpu <iq file> <buffer bytes (0 = whole file in one call)>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "libadsb_iface.hpp"

struct Listener : ADSB::IListener
{
    void OnChanged(ADSB::IAirCraft const& a) override
    {
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
