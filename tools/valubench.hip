// valubench.hip -- issue rate of the VALU ops scan1090 uses (profiling aid, not product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REP8(x) x x x x x x x x
#define KERNEL(name, asmtext)                                                                      \
    __global__ __launch_bounds__(256) void name(uint32_t* out, int iters)                           \
    {                                                                                                \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 ^ 11, a5 = a0 + 13, a6 = a0 * 17, a7 = a0 + 19; \
        uint32_t b = blockIdx.x + 3, c = threadIdx.x * 9 + 1;                                       \
        for (int i = 0; i < iters; i++)                                                              \
        {                                                                                            \
            REP8(asm volatile(asmtext : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) \
        }                                                                                            \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;              \
    }
#define OP8(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
#define OP8_3(op) op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n"
#define OP8_1(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
KERNEL(k_add_u32, OP8("v_add_u32"))
KERNEL(k_and_b32, OP8("v_and_b32"))
KERNEL(k_fma_f32, OP8_3("v_fma_f32"))
KERNEL(k_pk_max_u16, OP8("v_pk_max_u16"))
KERNEL(k_pk_sub_i16, OP8("v_pk_sub_i16"))
KERNEL(k_pk_mul_lo_u16, OP8("v_pk_mul_lo_u16"))
KERNEL(k_pk_mad_u16, OP8_3("v_pk_mad_u16"))
KERNEL(k_alignbit, OP8_3("v_alignbit_b32"))
KERNEL(k_perm, OP8_3("v_perm_b32"))
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %8, %9 bitop3:0x80\nv_bitop3_b32 %1, %1, %8, %9 bitop3:0x80\nv_bitop3_b32 %2, %2, %8, %9 bitop3:0x80\nv_bitop3_b32 %3, %3, %8, %9 bitop3:0x80\nv_bitop3_b32 %4, %4, %8, %9 bitop3:0x80\nv_bitop3_b32 %5, %5, %8, %9 bitop3:0x80\nv_bitop3_b32 %6, %6, %8, %9 bitop3:0x80\nv_bitop3_b32 %7, %7, %8, %9 bitop3:0x80\n")
KERNEL(k_lshl_or, OP8_3("v_lshl_or_b32"))
KERNEL(k_mul_u24, OP8("v_mul_u32_u24"))
KERNEL(k_mul_lo_u32, OP8("v_mul_lo_u32"))
KERNEL(k_sqrt_f32, OP8_1("v_sqrt_f32"))
KERNEL(k_cvt_f32_u32, OP8_1("v_cvt_f32_u32"))
KERNEL(k_max_u16, OP8("v_max_u16"))
KERNEL(k_max_u32, OP8("v_max_u32"))
KERNEL(k_sub_u16, OP8("v_sub_u16"))
KERNEL(k_max3_u32, OP8_3("v_max3_u32"))
KERNEL(k_sad_u16, OP8_3("v_sad_u16"))
KERNEL(k_dot2_u32_u16, OP8_3("v_dot2_u32_u16"))
KERNEL(k_dot4_i32_i8, OP8_3("v_dot4_i32_i8"))
KERNEL(k_pk_add_u16_sat, "v_pk_add_i16 %0, %0, %8 clamp\nv_pk_add_i16 %1, %1, %8 clamp\nv_pk_add_i16 %2, %2, %8 clamp\nv_pk_add_i16 %3, %3, %8 clamp\nv_pk_add_i16 %4, %4, %8 clamp\nv_pk_add_i16 %5, %5, %8 clamp\nv_pk_add_i16 %6, %6, %8 clamp\nv_pk_add_i16 %7, %7, %8 clamp\n")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_cmp_gt_u32, "v_cmp_gt_u32 vcc, %0, %8\nv_cmp_gt_u32 vcc, %1, %8\nv_cmp_gt_u32 vcc, %2, %8\nv_cmp_gt_u32 vcc, %3, %8\nv_cmp_gt_u32 vcc, %4, %8\nv_cmp_gt_u32 vcc, %5, %8\nv_cmp_gt_u32 vcc, %6, %8\nv_cmp_gt_u32 vcc, %7, %8\n")

template <typename K> void run(const char* name, K k, int ops_per_iter, int waves_per_simd)
{
    uint32_t* out; CK(hipMalloc(&out, 256 * 4096 * 4));
    const int iters = 2000, blocks = 256 * waves_per_simd; // 256-thread blocks: 4 waves = one per SIMD
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 10); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double instr_per_simd = (double)iters * ops_per_iter * waves_per_simd; // wave-instructions issued on each SIMD
    printf("%-18s waves/SIMD=%d  %7.3f ms  -> %.2f cycles per wave-instruction per SIMD @2.4GHz\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    CK(hipFree(out));
}
int main()
{
    for (int w : {1, 4})
    {
#define R(n) run(#n, n, 64, w)
        R(k_add_u32); R(k_and_b32); R(k_fma_f32); R(k_pk_max_u16); R(k_pk_sub_i16); R(k_pk_mul_lo_u16); R(k_pk_mad_u16); R(k_alignbit); R(k_perm);
        R(k_bitop3); R(k_lshl_or); R(k_mul_u24); R(k_mul_lo_u32); R(k_sqrt_f32); R(k_cvt_f32_u32); R(k_max_u16); R(k_max_u32); R(k_sub_u16);
        R(k_max3_u32); R(k_sad_u16); R(k_dot2_u32_u16); R(k_dot4_i32_i8); R(k_pk_add_u16_sat); R(k_mov_dpp); R(k_cmp_gt_u32);
    }
    return 0;
}
