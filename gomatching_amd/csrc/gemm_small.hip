// Tiny products of the tracker path: association logits  C[n_cur, N] = tgt . memory^T  (transformer.py:92-96 /
// lstmatcher.py:360-371) and similar GEMMs with at most a few thousand outputs and K up to a few thousand.  A 256x64
// MFMA tile spends 16 blocks x K/2 exact-fp32 MFMAs on such a product whatever M is (90 us at M=5, N=55, K=1024);
// and the skinny M <= 128 linear layers of the matcher transformers pay a two-kernel split-K (22 us) each.  Here one
// wave owns one output column for 8 rows on the VALU: latency of a few us, deterministic, independent of M.
#include "common.h"

namespace {

constexpr int RM = 8;                                                  // rows of A per wave

// One wave owns output column n for RM consecutive rows: the weight row streams once per wave (16-byte loads, 64
// lanes stride the K axis), the RM activation rows come from L1/L2, RM fp32 fmaf chains, fixed-order butterflies.
// An output's arithmetic depends only on (its row, its column, K): results do not change with M or N.
// NOTE (round 2): built WITHOUT packed-fp32 instructions like the whole library (build.py): as `v_pk_fma_f32` pairs these eight
// fmaf chains returned wrong LOW halves (= even rows) in 11-25 % of launches whenever waves of the bf16x6 GEMM kernel shared
// the SIMD -- the round-1 "tracker determinism" issue (tools/race_repro.py).
__global__ __launch_bounds__(256) void gemm_small_kernel(const float* __restrict__ A, const int* __restrict__ a_rows,
                                                         int lda, const float* __restrict__ W, int ldw,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ R, int ldr, int relu,
                                                         float* __restrict__ C, int ldc, int M, int N, int K) {
    const int lane = threadIdx.x & 63;
    const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // (row group, column), column fastest
    const int groups = (M + RM - 1) / RM;
    if (o >= (long)groups * N) return;
    const int n = (int)(o % N), m0 = (int)(o / N) * RM;
    const float* w = W + (size_t)n * ldw;
    const float* a[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        const int m = m0 + r < M ? m0 + r : M - 1;                     // clamp: tail rows recompute the last row
        a[r] = A + (size_t)(a_rows ? a_rows[m] : m) * lda;
    }
    float acc[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) acc[r] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 y = *reinterpret_cast<const f32x4*>(w + k);
#pragma unroll
        for (int r = 0; r < RM; ++r) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(a[r] + k);
            acc[r] = fmaf(x[0], y[0], acc[r]);
            acc[r] = fmaf(x[1], y[1], acc[r]);
            acc[r] = fmaf(x[2], y[2], acc[r]);
            acc[r] = fmaf(x[3], y[3], acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < RM; ++r) acc[r] = wave_sum(acc[r]);
    if (lane == 0) {
        const float sc = scale ? scale[n] : 1.f, sh = shift ? shift[n] : 0.f;
#pragma unroll
        for (int r = 0; r < RM; ++r) {
            const int m = m0 + r;
            if (m < M) {
                float v = acc[r] * sc + sh;
                if (R) v += R[(size_t)m * ldr + n];
                C[(size_t)m * ldc + n] = relu ? fmaxf(v, 0.f) : v;
            }
        }
    }
}

}  // namespace

extern "C" int gom_gemm_small_f32(const float* A, const int* a_rows, int lda, const float* W, int ldw,
                                  const float* scale, const float* shift, const float* R, int ldr, int relu, float* C,
                                  int ldc, int M, int N, int K, void* stream) {
    GOM_CHECK_ARG(A && W && C && M >= 0 && N > 0 && K > 0 && (K % 4) == 0);
    GOM_CHECK_ARG(lda >= K && ldw >= K && (lda % 4) == 0 && (ldw % 4) == 0 && ldc >= N && (!R || ldr >= N));
    GOM_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0);
    if (M == 0) return GOM_OK;
    GOM_CHECK_ARG((long)M * N <= (1L << 22));
    const long waves = (long)cdiv(M, RM) * N;
    hipLaunchKernelGGL(gemm_small_kernel, dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, (hipStream_t)stream, A, a_rows, lda,
                       W, ldw, scale, shift, R, ldr, relu, C, ldc, M, N, K);
    return gom_launch_status();
}
