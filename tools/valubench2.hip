// valubench2.hip -- issue cost (cycles per wave-instruction per SIMD) of the VALU ops a packed-16 / SWAR scan could use,
// at 1/2/4/8 waves per SIMD, plus two-op mixes (does a full-rate op hide beside a half-rate one?).  Profiling aid, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REP8(x) x x x x x x x x
#define KERNEL(name, asmtext)                                                                      \
    __global__ __launch_bounds__(256) void name(uint32_t* out, int iters)                           \
    {                                                                                                \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 ^ 11, a5 = a0 + 13, a6 = a0 * 17, a7 = a0 + 19; \
        uint32_t b = blockIdx.x + 3, c = threadIdx.x * 9 + 1;                                       \
        for (int i = 0; i < iters; i++)                                                              \
        {                                                                                            \
            REP8(asm volatile(asmtext : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");) \
        }                                                                                            \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;              \
    }
#define E2(op, i) op " %" #i ", %" #i ", %8\n"
#define E3(op, i) op " %" #i ", %" #i ", %8, %9\n"
#define E1(op, i) op " %" #i ", %" #i "\n"
#define OP8(op) E2(op,0) E2(op,1) E2(op,2) E2(op,3) E2(op,4) E2(op,5) E2(op,6) E2(op,7)
#define OP8_3(op) E3(op,0) E3(op,1) E3(op,2) E3(op,3) E3(op,4) E3(op,5) E3(op,6) E3(op,7)
#define OP8_1(op) E1(op,0) E1(op,1) E1(op,2) E1(op,3) E1(op,4) E1(op,5) E1(op,6) E1(op,7)
// suffix form: "op d, d, b <suffix>"
#define S2(op, i, suf) op " %" #i ", %" #i ", %8 " suf "\n"
#define OP8S(op, suf) S2(op,0,suf) S2(op,1,suf) S2(op,2,suf) S2(op,3,suf) S2(op,4,suf) S2(op,5,suf) S2(op,6,suf) S2(op,7,suf)
#define S3(op, i, suf) op " %" #i ", %" #i ", %8, %9 " suf "\n"
#define OP8S3(op, suf) S3(op,0,suf) S3(op,1,suf) S3(op,2,suf) S3(op,3,suf) S3(op,4,suf) S3(op,5,suf) S3(op,6,suf) S3(op,7,suf)
// mixes: 4 of A + 4 of B interleaved
#define MIX2(opa, opb) E2(opa,0) E2(opb,1) E2(opa,2) E2(opb,3) E2(opa,4) E2(opb,5) E2(opa,6) E2(opb,7)

KERNEL(k_add_u32, OP8("v_add_u32"))
KERNEL(k_sub_u32, OP8("v_sub_u32"))
KERNEL(k_and_b32, OP8("v_and_b32"))
KERNEL(k_xor_b32, OP8("v_xor_b32"))
KERNEL(k_lshlrev_b32, OP8("v_lshlrev_b32"))
KERNEL(k_lshrrev_b32, OP8("v_lshrrev_b32"))
KERNEL(k_max_u32, OP8("v_max_u32"))
KERNEL(k_min_u32, OP8("v_min_u32"))
KERNEL(k_max_i32, OP8("v_max_i32"))
KERNEL(k_max3_u32, OP8_3("v_max3_u32"))
KERNEL(k_min3_u32, OP8_3("v_min3_u32"))
KERNEL(k_med3_u32, OP8_3("v_med3_u32"))
KERNEL(k_add3_u32, OP8_3("v_add3_u32"))
KERNEL(k_lshl_add_u32, OP8_3("v_lshl_add_u32"))
KERNEL(k_lshl_or_b32, OP8_3("v_lshl_or_b32"))
KERNEL(k_and_or_b32, OP8_3("v_and_or_b32"))
KERNEL(k_or3_b32, OP8_3("v_or3_b32"))
KERNEL(k_bfi_b32, OP8_3("v_bfi_b32"))
KERNEL(k_bfe_u32, OP8_3("v_bfe_u32"))
KERNEL(k_bitop3_b32, OP8S3("v_bitop3_b32", "bitop3:0x80"))
KERNEL(k_alignbit_b32, OP8_3("v_alignbit_b32"))
KERNEL(k_alignbyte_b32, OP8_3("v_alignbyte_b32"))
KERNEL(k_perm_b32, OP8_3("v_perm_b32"))
KERNEL(k_cndmask_b32, "v_cmp_gt_u32 vcc, %8, %9\n" OP8S("v_cndmask_b32", ", vcc"))
KERNEL(k_mul_u32_u24, OP8("v_mul_u32_u24"))
KERNEL(k_mad_u32_u24, OP8_3("v_mad_u32_u24"))
KERNEL(k_mul_lo_u32, OP8("v_mul_lo_u32"))
KERNEL(k_mad_u32_u16, OP8_3("v_mad_u32_u16"))
KERNEL(k_sad_u8, OP8_3("v_sad_u8"))
KERNEL(k_sad_u16, OP8_3("v_sad_u16"))
KERNEL(k_sad_u32, OP8_3("v_sad_u32"))
KERNEL(k_msad_u8, OP8_3("v_msad_u8"))
KERNEL(k_lerp_u8, OP8_3("v_lerp_u8"))
KERNEL(k_dot2_u32_u16, OP8_3("v_dot2_u32_u16"))
KERNEL(k_dot2_i32_i16, OP8_3("v_dot2_i32_i16"))
KERNEL(k_dot4_i32_i8, OP8_3("v_dot4_i32_i8"))
KERNEL(k_dot4_u32_u8, OP8_3("v_dot4_u32_u8"))
KERNEL(k_dot8_u32_u4, OP8_3("v_dot8_u32_u4"))
KERNEL(k_pk_max_u16, OP8("v_pk_max_u16"))
KERNEL(k_pk_min_u16, OP8("v_pk_min_u16"))
KERNEL(k_pk_sub_i16, OP8("v_pk_sub_i16"))
KERNEL(k_pk_add_u16, OP8("v_pk_add_u16"))
KERNEL(k_pk_mul_lo_u16, OP8("v_pk_mul_lo_u16"))
KERNEL(k_pk_mad_u16, OP8_3("v_pk_mad_u16"))
KERNEL(k_pk_lshrrev_b16, OP8("v_pk_lshrrev_b16"))
KERNEL(k_pk_ashrrev_i16, OP8("v_pk_ashrrev_i16"))
KERNEL(k_pk_max_f16, OP8("v_pk_max_f16"))
KERNEL(k_pk_add_f16, OP8("v_pk_add_f16"))
KERNEL(k_pk_fma_f16, OP8_3("v_pk_fma_f16"))
KERNEL(k_pk_maximum3_f16, OP8_3("v_pk_maximum3_f16"))
KERNEL(k_pk_minimum3_f16, OP8_3("v_pk_minimum3_f16"))
KERNEL(k_max_u16, OP8("v_max_u16"))
KERNEL(k_sub_u16, OP8("v_sub_u16"))
KERNEL(k_mad_u16, OP8_3("v_mad_u16"))
KERNEL(k_max_f32, OP8("v_max_f32"))
KERNEL(k_min_f32, OP8("v_min_f32"))
KERNEL(k_sub_f32, OP8("v_sub_f32"))
KERNEL(k_fma_f32, OP8_3("v_fma_f32"))
KERNEL(k_max3_f32, OP8_3("v_max3_f32"))
KERNEL(k_med3_f32, OP8_3("v_med3_f32"))
KERNEL(k_maximum3_f32, OP8_3("v_maximum3_f32"))
KERNEL(k_sqrt_f32, OP8_1("v_sqrt_f32"))
KERNEL(k_cvt_f32_u32, OP8_1("v_cvt_f32_u32"))
KERNEL(k_cvt_f32_ubyte1, OP8_1("v_cvt_f32_ubyte1"))
KERNEL(k_cvt_pk_u16_u32, OP8("v_cvt_pk_u16_u32"))
KERNEL(k_mov_b32, OP8_1("v_mov_b32"))
KERNEL(k_mov_dpp_rowshr, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_add_dpp_rowshr, "v_add_u32_dpp %0, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %1, %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %2, %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %3, %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %4, %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %5, %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %6, %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %7, %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_sub_sdwa_w1, OP8S("v_sub_u32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0"))
KERNEL(k_max_u16_sdwa_hi, OP8S("v_max_u16_sdwa", "dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1"))
KERNEL(k_cmp_gt_u32, "v_cmp_gt_u32 vcc, %0, %8\nv_cmp_gt_u32 vcc, %1, %8\nv_cmp_gt_u32 vcc, %2, %8\nv_cmp_gt_u32 vcc, %3, %8\nv_cmp_gt_u32 vcc, %4, %8\nv_cmp_gt_u32 vcc, %5, %8\nv_cmp_gt_u32 vcc, %6, %8\nv_cmp_gt_u32 vcc, %7, %8\n")
KERNEL(k_cmp_gt_u16, "v_cmp_gt_u16 vcc, %0, %8\nv_cmp_gt_u16 vcc, %1, %8\nv_cmp_gt_u16 vcc, %2, %8\nv_cmp_gt_u16 vcc, %3, %8\nv_cmp_gt_u16 vcc, %4, %8\nv_cmp_gt_u16 vcc, %5, %8\nv_cmp_gt_u16 vcc, %6, %8\nv_cmp_gt_u16 vcc, %7, %8\n")
KERNEL(k_cmp_gt_f32, "v_cmp_gt_f32 vcc, %0, %8\nv_cmp_gt_f32 vcc, %1, %8\nv_cmp_gt_f32 vcc, %2, %8\nv_cmp_gt_f32 vcc, %3, %8\nv_cmp_gt_f32 vcc, %4, %8\nv_cmp_gt_f32 vcc, %5, %8\nv_cmp_gt_f32 vcc, %6, %8\nv_cmp_gt_f32 vcc, %7, %8\n")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3\nv_readlane_b32 s21, %1, 3\nv_readlane_b32 s22, %2, 3\nv_readlane_b32 s23, %3, 3\nv_readlane_b32 s20, %4, 3\nv_readlane_b32 s21, %5, 3\nv_readlane_b32 s22, %6, 3\nv_readlane_b32 s23, %7, 3\n")
KERNEL(k_permlane32_swap, "v_permlane32_swap_b32 %0, %1\nv_permlane32_swap_b32 %2, %3\nv_permlane32_swap_b32 %4, %5\nv_permlane32_swap_b32 %6, %7\nv_permlane32_swap_b32 %0, %1\nv_permlane32_swap_b32 %2, %3\nv_permlane32_swap_b32 %4, %5\nv_permlane32_swap_b32 %6, %7\n")
KERNEL(k_mix_pkmax_add, MIX2("v_pk_max_u16", "v_add_u32"))
KERNEL(k_mix_pkmax_and, MIX2("v_pk_max_u16", "v_and_b32"))
KERNEL(k_mix_pksub_pkmax, MIX2("v_pk_sub_i16", "v_pk_max_u16"))
KERNEL(k_mix_add_and, MIX2("v_add_u32", "v_and_b32"))
KERNEL(k_mix_maxu32_add, MIX2("v_max_u32", "v_add_u32"))

struct Entry { const char* name; void (*k)(uint32_t*, int); };
#define R(n) {#n, n}
static Entry entries[] = {
    R(k_add_u32), R(k_sub_u32), R(k_and_b32), R(k_xor_b32), R(k_lshlrev_b32), R(k_lshrrev_b32), R(k_max_u32), R(k_min_u32), R(k_max_i32), R(k_max3_u32), R(k_min3_u32),
    R(k_med3_u32), R(k_add3_u32), R(k_lshl_add_u32), R(k_lshl_or_b32), R(k_and_or_b32), R(k_or3_b32), R(k_bfi_b32), R(k_bfe_u32), R(k_bitop3_b32), R(k_alignbit_b32),
    R(k_alignbyte_b32), R(k_perm_b32), R(k_cndmask_b32), R(k_mul_u32_u24), R(k_mad_u32_u24), R(k_mul_lo_u32), R(k_mad_u32_u16), R(k_sad_u8), R(k_sad_u16), R(k_sad_u32),
    R(k_msad_u8), R(k_lerp_u8), R(k_dot2_u32_u16), R(k_dot2_i32_i16), R(k_dot4_i32_i8), R(k_dot4_u32_u8), R(k_dot8_u32_u4), R(k_pk_max_u16), R(k_pk_min_u16), R(k_pk_sub_i16),
    R(k_pk_add_u16), R(k_pk_mul_lo_u16), R(k_pk_mad_u16), R(k_pk_lshrrev_b16), R(k_pk_ashrrev_i16), R(k_pk_max_f16), R(k_pk_add_f16), R(k_pk_fma_f16), R(k_pk_maximum3_f16),
    R(k_pk_minimum3_f16), R(k_max_u16), R(k_sub_u16), R(k_mad_u16), R(k_max_f32), R(k_min_f32), R(k_sub_f32), R(k_fma_f32), R(k_max3_f32), R(k_med3_f32), R(k_maximum3_f32),
    R(k_sqrt_f32), R(k_cvt_f32_u32), R(k_cvt_f32_ubyte1), R(k_cvt_pk_u16_u32), R(k_mov_b32), R(k_mov_dpp_rowshr), R(k_add_dpp_rowshr), R(k_sub_sdwa_w1), R(k_max_u16_sdwa_hi),
    R(k_cmp_gt_u32), R(k_cmp_gt_u16), R(k_cmp_gt_f32), R(k_readlane), R(k_permlane32_swap),
    R(k_mix_pkmax_add), R(k_mix_pkmax_and), R(k_mix_pksub_pkmax), R(k_mix_add_and), R(k_mix_maxu32_add),
};

static double run(void (*k)(uint32_t*, int), int waves_per_simd, uint32_t* out)
{
    const int iters = 1000, blocks = 256 * waves_per_simd; // 256-thread blocks: 4 waves = one per SIMD
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 10); CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 3; r++)
    {
        CK(hipEventRecord(a)); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    double instr_per_simd = (double)iters * 64 * waves_per_simd;
    return best * 1e-3 * 2.4e9 / instr_per_simd;
}
int main()
{
    uint32_t* out; CK(hipMalloc(&out, 256 * 256 * 8 * 4));
    printf("cycles per wave-instruction per SIMD at 2.4 GHz nominal (the chip may clock lower under load)\n%-22s %8s %8s %8s %8s\n", "op", "1 w/SIMD", "2", "4", "8");
    for (auto& e : entries)
    {
        printf("%-22s", e.name);
        for (int w : {1, 2, 4, 8}) printf(" %8.2f", run(e.k, w, out));
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
