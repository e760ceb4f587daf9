// membench.hip -- what does this box deliver for the access shapes scan1090 could use? (profiling aid, not product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// K0: classic grid-stride 16 B/lane read, 256-thread blocks
__global__ __launch_bounds__(256) void k0(const uint4* __restrict__ p, size_t n16, uint32_t* out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n16; i += st) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
// K1: persistent single-wave workgroups, each iteration reads one contiguous 8 KiB chunk (8 x 1 KiB rows) + 512 B halo
template <int WG, bool LDS, bool BAR>
__global__ __launch_bounds__(WG) void k1(const uint8_t* __restrict__ base, uint32_t nchunks, uint32_t* out)
{
    __shared__ uint4 tile[LDS ? 9 * 64 * (WG / 64) : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t gw = blockIdx.x * (WG / 64) + wave, nw = gridDim.x * (WG / 64);
    uint32_t acc = 0;
    for (uint32_t c = gw; c < nchunks; c += nw)
    {
        const uint4* p = reinterpret_cast<const uint4*>(base + (size_t)c * 8192) + lane;
        uint4 r[9];
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = p[k * 64];
        r[8] = (lane < 32 && c + 1 < nchunks) ? p[8 * 64] : make_uint4(0, 0, 0, 0);
        if (BAR) __syncthreads();
#pragma unroll
        for (int k = 0; k < 9; k++)
        {
            if (LDS) tile[(wave * 9 + k) * 64 + lane] = r[k];
            acc ^= r[k].x ^ r[k].y ^ r[k].z ^ r[k].w;
        }
        if (BAR) __syncthreads();
        if (LDS) acc ^= tile[(wave * 9 + 3) * 64 + (lane ^ 1)].x;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <typename F> float timeit(F f, int reps = 10)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main()
{
    const size_t bytes = 1ull << 30;
    uint8_t* d; uint32_t* out; CK(hipMalloc(&d, bytes + 4096)); CK(hipMalloc(&out, 64));
    CK(hipMemset(d, 0x7f, bytes + 4096));
    const uint32_t nchunks = bytes / 8192;
    auto rep = [&](const char* name, float ms) { printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6); };
    rep("k0 grid-stride 256thr x 2048 blocks", timeit([&] { hipLaunchKernelGGL(k0, dim3(2048), dim3(256), 0, 0, (const uint4*)d, bytes / 16, out); }));
    rep("k0 grid-stride 256thr x 8192 blocks", timeit([&] { hipLaunchKernelGGL(k0, dim3(8192), dim3(256), 0, 0, (const uint4*)d, bytes / 16, out); }));
    for (int g : {2048, 4096, 8192, 16384})
    {
        char nm[96];
        snprintf(nm, sizeof nm, "k1 wg64  grid %5d  noLDS nobar", g); rep(nm, timeit([&] { hipLaunchKernelGGL((k1<64, false, false>), dim3(g), dim3(64), 0, 0, d, nchunks, out); }));
        snprintf(nm, sizeof nm, "k1 wg64  grid %5d  LDS   bar  ", g); rep(nm, timeit([&] { hipLaunchKernelGGL((k1<64, true, true>), dim3(g), dim3(64), 0, 0, d, nchunks, out); }));
    }
    for (int g : {1024, 2048, 4096})
    {
        char nm[96];
        snprintf(nm, sizeof nm, "k1 wg256 grid %5d  noLDS nobar", g); rep(nm, timeit([&] { hipLaunchKernelGGL((k1<256, false, false>), dim3(g), dim3(256), 0, 0, d, nchunks, out); }));
        snprintf(nm, sizeof nm, "k1 wg256 grid %5d  LDS   bar  ", g); rep(nm, timeit([&] { hipLaunchKernelGGL((k1<256, true, true>), dim3(g), dim3(256), 0, 0, d, nchunks, out); }));
    }
    return 0;
}
