// salubench.hip -- do SALU instructions steal issue slots from VALU on gfx950? (profiling aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REP8(x) x x x x x x x x
// MODE 0: 64 VALU (v_add_u32) per iter; 1: 64 SALU (s_add_u32); 2: 64 VALU + 64 SALU interleaved; 3: 64 pk VALU; 4: 64 pk VALU + 64 SALU
// 5: 64 s_and_b64; 6: 64 VALU + 64 s_and_b64
template <int MODE> __global__ __launch_bounds__(256) void k(uint32_t* out, int iters)
{
    uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = blockIdx.x + 3;
    uint32_t s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    uint64_t m0 = blockIdx.x, m1 = 77;
    for (int i = 0; i < iters; i++)
    {
        if (MODE == 0) { REP8(REP8(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(b));)) }
        if (MODE == 3) { REP8(REP8(asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a0) : "v"(b));)) }
        if (MODE == 1) { REP8(REP8(asm volatile("s_add_u32 %0, %0, 3" : "+s"(s0) : : "scc");)) }
        if (MODE == 5) { REP8(REP8(asm volatile("s_and_b64 %0, %0, %1" : "+s"(m0) : "s"(m1) : "scc");)) }
        if (MODE == 2) { REP8(REP8(asm volatile("v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, 3" : "+v"(a0), "+s"(s0) : "v"(b) : "scc");)) }
        if (MODE == 4) { REP8(REP8(asm volatile("v_pk_max_u16 %0, %0, %2\n s_add_u32 %1, %1, 3" : "+v"(a0), "+s"(s0) : "v"(b) : "scc");)) }
        if (MODE == 6) { REP8(REP8(asm volatile("v_add_u32 %0, %0, %2\n s_and_b64 %1, %1, %3" : "+v"(a0), "+s"(m0) : "v"(b), "s"(m1) : "scc");)) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ s0 ^ s1 ^ s2 ^ s3 ^ (uint32_t)m0;
}
template <int MODE> void run(const char* name, int w)
{
    uint32_t* out; CK(hipMalloc(&out, 256 * 4096 * 4));
    const int iters = 500, blocks = 256 * w;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-34s waves/SIMD=%d %8.3f ms -> %.2f cycles per 'slot' (64 per iter) per SIMD @2.4GHz\n", name, w, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * w));
    CK(hipFree(out));
}
int main()
{
    for (int w : {1, 2, 4})
    {
        run<0>("64 v_add_u32 (dependent chain)", w); run<3>("64 v_pk_max_u16 (dep chain)", w); run<1>("64 s_add_u32 (dep chain)", w); run<5>("64 s_and_b64 (dep chain)", w);
        run<2>("64 x (v_add_u32 + s_add_u32)", w); run<4>("64 x (v_pk_max_u16 + s_add_u32)", w); run<6>("64 x (v_add_u32 + s_and_b64)", w);
    }
    return 0;
}
