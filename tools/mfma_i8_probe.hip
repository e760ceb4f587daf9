// tools/mfma_i8_probe.hip -- what the 2.4 MS/s gate needs to know about v_mfma_i32_32x32x32_i8 on gfx950 before it is built on it:
//   (1) the operand and result lane maps (exact integer data, asymmetric operands);
//   (2) what an MFMA costs a wave that is otherwise bound by vector-instruction issue: 16 waves per CU (the scan kernels' occupancy), each
//       running blocks of V packed 16-bit additions with and without M MFMAs per block.
// hipcc --offload-arch=gfx950 -O3 -o mfma_i8_probe tools/mfma_i8_probe.hip && ./mfma_i8_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                                   \
    do                                                                                             \
    {                                                                                              \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess)                                                                      \
        {                                                                                          \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                                         \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

__global__ void layout_kernel(const v4i* a, const v4i* b, v16i* d)
{
    const int l = threadIdx.x;
    v16i      c = {};
    c           = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[l], b[l], c, 0, 0, 0);
    d[l]        = c;
}

// V dependent-free packed adds on 8 registers + M MFMAs (two accumulator chains), `iters` times
template <int V, int M>
__global__ __launch_bounds__(64, 4) void mix_kernel(uint32_t* out, int iters, v4i a, v4i b)
{
    uint32_t r[8];
    for (int k = 0; k < 8; k++) r[k] = threadIdx.x * 0x01010101u + k;
    v16i acc0 = {}, acc1 = {};
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int m = 0; m < M; m++)
        {
            if (m & 1) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < V; v++) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[v & 7]) : "v"(r[(v + 3) & 7]));
    }
    uint32_t x = 0;
    for (int k = 0; k < 8; k++) x ^= r[k];
    for (int k = 0; k < 16; k++) x ^= (uint32_t)(acc0[k] ^ acc1[k]);
    if (x == 0x12345678u) out[0] = x;
}

template <int V, int M>
static float time_mix(int iters, uint32_t* d_out)
{
    v4i a = {0x01020304, 0x01010101, 0x02020202, 0x01010101}, b = {0x01010101, 0x01020102, 0x01010101, 0x03010301};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((mix_kernel<V, M>), dim3(4096), dim3(64), 0, 0, d_out, iters, a, b); // warm
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++)
    {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((mix_kernel<V, M>), dim3(4096), dim3(64), 0, 0, d_out, iters, a, b);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    return best;
}

int main()
{
    // ---- (1) lane maps
    int8_t A[32][32], B[32][32]; // A[row][k], B[k][col]
    srand(1);
    for (int i = 0; i < 32; i++)
        for (int k = 0; k < 32; k++) A[i][k] = (int8_t)(rand() % 17 - 8), B[i][k] = (int8_t)(rand() % 19 - 9);
    int ref[32][32];
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++)
        {
            int s = 0;
            for (int k = 0; k < 32; k++) s += (int)A[i][k] * (int)B[k][j];
            ref[i][j] = s;
        }
    for (int hyp = 0; hyp < 2; hyp++)
    {
        // hyp 0: byte b of lane l is k = 16 (l / 32) + b.  hyp 1: k = 8 (l / 32) + (b % 8) + 16 (b / 8)
        std::vector<int8_t> ha(64 * 16), hb(64 * 16);
        for (int l = 0; l < 64; l++)
            for (int bb = 0; bb < 16; bb++)
            {
                const int k     = hyp == 0 ? 16 * (l / 32) + bb : 8 * (l / 32) + (bb % 8) + 16 * (bb / 8);
                ha[l * 16 + bb] = A[l % 32][k];
                hb[l * 16 + bb] = B[k][l % 32];
            }
        v4i * da, *db;
        v16i* dd;
        CHECK(hipMalloc(&da, 64 * 16));
        CHECK(hipMalloc(&db, 64 * 16));
        CHECK(hipMalloc(&dd, 64 * 64));
        CHECK(hipMemcpy(da, ha.data(), 64 * 16, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(db, hb.data(), 64 * 16, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, da, db, dd);
        std::vector<int> hd(64 * 16);
        CHECK(hipMemcpy(hd.data(), dd, 64 * 64, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < 16; r++)
            {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
                bad += hd[l * 16 + r] != ref[row][col];
            }
        printf("layout: operand k-map hypothesis %d (%s), result map col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): %d of 1024 wrong\n", hyp,
               hyp == 0 ? "byte b of lane l: k = 16 (l >> 5) + b" : "k = 8 (l >> 5) + (b & 7) + 16 (b >> 3)", bad);
    }
    // both maps give the same product if A and B use the SAME map (the sum over k does not care about the order): shown by hypothesis 0 and 1 both passing or not

    // ---- (2) MFMA beside vector-issue-bound work, 16 waves per CU
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, 64));
    const int iters = 2000;
    printf("mix: 4096 waves (16 per CU), %d iterations of [M x v_mfma_i32_32x32x32_i8 + V x v_pk_add_u16]; ms per launch and ns per iteration per wave-slot\n", iters);
#define ROW(V, M)                                                                                                     \
    {                                                                                                                 \
        const float ms = time_mix<V, M>(iters, d_out);                                                                \
        printf("  V = %3d  M = %d : %8.4f ms   %7.1f ns per iteration\n", V, M, ms, ms * 1e6f / iters);               \
    }
    ROW(0, 1) ROW(0, 2) ROW(0, 4) ROW(64, 0) ROW(64, 1) ROW(64, 2) ROW(64, 4) ROW(128, 0) ROW(128, 1) ROW(128, 2) ROW(128, 4) ROW(256, 0) ROW(256, 2) ROW(256, 4) ROW(256, 8)
    return 0;
}
