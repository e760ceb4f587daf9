// tools/uatbench.hip -- times the fused UAT scan kernel alone on random IQ (variants via -D macros in uat978.hip)
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -Iinclude -Ilibadsb_amd/csrc tools/uatbench.hip -o /tmp/uatbench [-DUAT_VARIANT=n]
#include "../libadsb_amd/csrc/uat978.hip"

#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv)
{
    const size_t mib = argc > 1 ? atoi(argv[1]) : 1024;
    const size_t n   = mib * (1 << 20) / 2;
    const int    mode = argc > 2 ? atoi(argv[2]) : 0; // 0 quiet noise, 1 uniform random IQ, 2 ring of radius ~60 (a carrier)
    std::vector<uint16_t> lut(65536);
    for (int i = 0; i < 65536; i++) lut[i] = (uint16_t)(i * 40503u >> 3);
    std::vector<uint16_t> h(n);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < n; i++)
    {
        x ^= x << 13, x ^= x >> 7, x ^= x << 17;
        if (mode == 0) h[i] = (uint16_t)((127 + (int)(x & 7) - 3) | ((127 + (int)((x >> 8) & 7) - 3) << 8)); // quiet noise around the centre
        else if (mode == 1) h[i] = (uint16_t)x;
        else
        {
            const double ang = (double)(i % 977) * 0.6 + (double)(x & 0xFFFF) * 1e-5;
            h[i] = (uint16_t)((int)(127.5 + 60 * cos(ang)) | ((int)(127.5 + 60 * sin(ang)) << 8));
        }
    }
    uint16_t *d, *l;
    uint32_t *cand, *counts;
    hipMalloc(&d, n * 2), hipMalloc(&l, 131072), hipMalloc(&cand, 4 << 20), hipMalloc(&counts, 8);
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice), hipMemcpy(l, lut.data(), 131072, hipMemcpyHostToDevice);
    adsb_amd::UatArgs a{};
    a.in = d, a.lut = l, a.nsamples = n, a.phases_given = 0, a.cand = cand, a.cand_cap = 1 << 20, a.counts = counts;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float best = 1e9;
    const int idle_ms = argc > 3 ? atoi(argv[3]) : 0; // host pause before each launch (lets the GPU drop its clocks)
    for (int it = 0; it < 6; it++)
    {
        if (idle_ms) usleep(idle_ms * 1000);
        hipEventRecord(e0, 0);
        adsb_amd::launch_uat978(a, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        if (idle_ms) printf("  it %d: %.4f ms\n", it, ms);
    }
    uint32_t c[2];
    hipMemcpy(c, counts, 8, hipMemcpyDeviceToHost);
    printf("%zu MiB: %.4f ms  %.1f GB/s  candidates %u\n", mib, best, n * 2 / best / 1e6, c[0]);
    return 0;
}
