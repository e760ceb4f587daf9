// test_978_gpu.cpp -- the reference's UAT flow against the GPU handler: the handler comes from
// ADSB::test::TryCreateUAT978Handler, buffers go through HandleData, and this binary plays the host's part of the seam by
// defining dump_raw_message (uat2json-wrapper.cpp:14), which prints one line per frame.  The Python test compares the lines
// with the oracle's.   usage: test_978_gpu <iq file> <bytes per HandleData call>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "libadsb_iface.hpp"

static ADSB::TrafficManager* g_expected_manager = nullptr;

extern "C" void dump_raw_message(char updown, uint8_t* data, int len, int rs_errors)
{
    // the reference's up-call finds its traffic manager through the thread-local slot HandleData fills (UAT978.cpp:46)
    if (*ADSB::GetThreadLocalTrafficManager() != g_expected_manager) std::printf("!! thread-local traffic manager not set\n");
    std::printf("%c %d %d ", updown, len, rs_errors);
    for (int i = 0; i < len; i++) std::printf("%02x", data[i]);
    std::printf("\n");
}

struct Selector : RTLSDR::IDeviceSelector
{
    [[nodiscard]] bool SelectDevice(RTLSDR::DeviceInfo const&) const override { return false; }
};

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    std::ifstream        f(argv[1], std::ios::binary);
    std::vector<uint8_t> iq((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const size_t         step = std::strtoull(argv[2], nullptr, 10);
    auto                 mgr  = std::make_shared<ADSB::TrafficManager>();
    g_expected_manager        = mgr.get();
    Selector selector;
    auto     handler = ADSB::test::TryCreateUAT978Handler(mgr, &selector, ADSB::Source::UAT978);
    for (size_t o = 0; o < iq.size(); o += step) handler->HandleData({iq.data() + o, std::min(step, iq.size() - o)});
    return 0;
}
