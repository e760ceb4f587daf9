// valubench3.hip -- do full-rate (FR, 2.5 cycles) VALU ops keep their rate when other waves of the same SIMD issue half-rate
// (HR, 4.2 cycles) ops, and when one wave alternates long FR and HR runs?  One 1024-thread workgroup per CU = 4 waves per SIMD
// (waves w, w+4, w+8, w+12 share a SIMD).  Profiling aid, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define E2(op, i) op " %" #i ", %" #i ", %8\n"
#define OP8(op) E2(op,0) E2(op,1) E2(op,2) E2(op,3) E2(op,4) E2(op,5) E2(op,6) E2(op,7)
#define REP4(x) x x x x
#define REP8(x) x x x x x x x x
#define ASM(text) asm volatile(text : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#define FR64 REP8(ASM(OP8("v_sub_u32")))
#define HR64 REP8(ASM(OP8("v_pk_max_u16")))
#define FR32 REP4(ASM(OP8("v_sub_u32")))
#define HR32 REP4(ASM(OP8("v_pk_max_u16")))
#define FR8 ASM(OP8("v_sub_u32"))
#define HR8 ASM(OP8("v_pk_max_u16"))
#define HEAD uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 ^ 11, a5 = a0 + 13, a6 = a0 * 17, a7 = a0 + 19; \
             uint32_t b = blockIdx.x + 3, c = threadIdx.x * 9 + 1; const int wave = threadIdx.x >> 6; (void)wave; (void)c;
#define TAIL out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
// mode 0: every wave FR.  1: every wave HR.  2: two waves of each SIMD FR, two HR (wave >> 2 odd -> HR).
// 3: every wave alternates 64 FR / 64 HR.  4: alternates 8 FR / 8 HR.  5: like 3 but waves (wave>>2) odd start with HR (anti-phase).
// 6: three of the four waves of a SIMD FR, one HR.  7: one FR, three HR.
__global__ __launch_bounds__(1024) void k(uint32_t* out, int iters, int mode)
{
    HEAD
    const bool odd = (wave >> 2) & 1;
    const int  q   = wave >> 2;
    if (mode == 0) for (int i = 0; i < iters; i++) { FR64 FR64 }
    else if (mode == 1) for (int i = 0; i < iters; i++) { HR64 HR64 }
    else if (mode == 2) { if (odd) for (int i = 0; i < iters; i++) { HR64 HR64 } else for (int i = 0; i < iters; i++) { FR64 FR64 } }
    else if (mode == 3) for (int i = 0; i < iters; i++) { FR64 HR64 }
    else if (mode == 4) for (int i = 0; i < iters; i++) { FR8 HR8 FR8 HR8 FR8 HR8 FR8 HR8 FR8 HR8 FR8 HR8 FR8 HR8 FR8 HR8 }
    else if (mode == 5) { if (odd) for (int i = 0; i < iters; i++) { HR64 FR64 } else for (int i = 0; i < iters; i++) { FR64 HR64 } }
    else if (mode == 6) { if (q == 3) for (int i = 0; i < iters; i++) { HR64 HR64 } else for (int i = 0; i < iters; i++) { FR64 FR64 } }
    else if (mode == 7) { if (q != 3) for (int i = 0; i < iters; i++) { HR64 HR64 } else for (int i = 0; i < iters; i++) { FR64 FR64 } }
    TAIL
}
int main()
{
    uint32_t* out; CK(hipMalloc(&out, 256 * 1024 * 4));
    const int iters = 2000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char* what[] = {"all FR", "all HR", "2 FR + 2 HR waves per SIMD", "each wave 64 FR / 64 HR", "each wave 8 FR / 8 HR", "64/64, half the waves in anti-phase",
                          "3 FR + 1 HR waves per SIMD", "1 FR + 3 HR waves per SIMD"};
    for (int mode = 0; mode < 8; mode++)
    {
        hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, out, 10, mode); CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int r = 0; r < 3; r++)
        {
            CK(hipEventRecord(a)); hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, out, iters, mode); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        // 4 waves per SIMD x 128 instructions per iteration
        printf("mode %d %-40s %8.3f ms  = %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", mode, what[mode], best, best * 1e-3 * 2.4e9 / (4.0 * 128 * iters));
    }
    return 0;
}
