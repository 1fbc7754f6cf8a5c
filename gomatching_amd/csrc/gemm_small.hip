// Tiny products of the tracker path: association logits  C[n_cur, N] = tgt . memory^T  (transformer.py:92-96 /
// lstmatcher.py:360-371) and similar GEMMs with at most a few thousand outputs and K up to a few thousand.  A 256x64
// MFMA tile spends 16 blocks x K/2 exact-fp32 MFMAs on such a product whatever M is (90 us at M=5, N=55, K=1024);
// and the skinny M <= 128 linear layers of the matcher transformers pay a two-kernel split-K (22 us) each.  Here one
// wave owns one output column for 8 rows on the VALU: latency of a few us, deterministic, independent of M.
#include "common.h"

namespace {

constexpr int RM = 8;                                                  // rows of A per wave
constexpr int CN = 8;                                                  // output columns per wave

// One wave owns an RM x CN patch of outputs: per 256-wide k-step CN weight quads and RM activation quads feed RM x CN fmaf
// chains (1 KB of loads per output instead of 4.5 with one column per wave: the kernel was bound by re-reading A through
// L1), 64 lanes stride the K axis.  The 64 per-lane partial sums are reduced by a TRANSPOSING butterfly: at offset o a lane
// keeps the half of its values whose index has bit o equal to its own lane bit and adds the partner's copy of that half --
// 63 exchanges instead of 64 x 6, the same (own + partner) tree at offsets 32, 16, ..., 1 as a per-value wave_sum, so every
// output has exactly the bits the one-column kernel gave it; lane l ends up with output (column l >> 3, row l & 7).
// An output's arithmetic depends only on (its row, its column, K): results do not change with M or N.
// NOTE (round 2): built WITHOUT packed-fp32 instructions like the whole library (build.py): as `v_pk_fma_f32` pairs such
// adjacent fmaf chains returned wrong LOW halves (= even rows) in 11-25 % of launches whenever waves of the bf16x6 GEMM kernel
// shared the SIMD -- the round-1 "tracker determinism" issue (tools/race_repro.py).
__global__ __launch_bounds__(256) void gemm_small_kernel(const float* __restrict__ A, const int* __restrict__ a_rows,
                                                         int lda, const float* __restrict__ W, int ldw,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ R, int ldr, int relu,
                                                         float* __restrict__ C, int ldc, int M, int N, int K) {
    const int lane = threadIdx.x & 63;
    const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // (row group, column group), column group fastest
    const int row_groups = (M + RM - 1) / RM, col_groups = (N + CN - 1) / CN;
    if (o >= (long)row_groups * col_groups) return;
    const int n0 = (int)(o % col_groups) * CN, m0 = (int)(o / col_groups) * RM;
    const float* w[CN];
    const float* a[RM];
#pragma unroll
    for (int c = 0; c < CN; ++c) w[c] = W + (size_t)(n0 + c < N ? n0 + c : N - 1) * ldw;   // clamp: tail patches recompute
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        const int m = m0 + r < M ? m0 + r : M - 1;
        a[r] = A + (size_t)(a_rows ? a_rows[m] : m) * lda;
    }
    float v[CN * RM];                                                  // index c * RM + r
#pragma unroll
    for (int j = 0; j < CN * RM; ++j) v[j] = 0.f;
#pragma unroll 2
    for (int k = lane * 4; k < K; k += 256) {
        f32x4 y[CN], x[RM];
#pragma unroll
        for (int c = 0; c < CN; ++c) y[c] = *reinterpret_cast<const f32x4*>(w[c] + k);
#pragma unroll
        for (int r = 0; r < RM; ++r) x[r] = *reinterpret_cast<const f32x4*>(a[r] + k);
#pragma unroll
        for (int c = 0; c < CN; ++c)
#pragma unroll
            for (int r = 0; r < RM; ++r) {
                float t = v[c * RM + r];
                t = fmaf(x[r][0], y[c][0], t);
                t = fmaf(x[r][1], y[c][1], t);
                t = fmaf(x[r][2], y[c][2], t);
                t = fmaf(x[r][3], y[c][3], t);
                v[c * RM + r] = t;
            }
    }
#pragma unroll
    for (int half = CN * RM / 2; half > 0; half >>= 1) {
        const bool up = (lane & half) != 0;
#pragma unroll
        for (int j = 0; j < half; ++j) {
            const float send = up ? v[j] : v[j + half];
            const float keep = up ? v[j + half] : v[j];
            v[j] = keep + __shfl_xor(send, half, 64);
        }
    }
    const int n = n0 + (lane >> 3), m = m0 + (lane & 7);
    if (n < N && m < M) {
        const float sc = scale ? scale[n] : 1.f, sh = shift ? shift[n] : 0.f;
        float y = v[0] * sc + sh;
        if (R) y += R[(size_t)m * ldr + n];
        C[(size_t)m * ldc + n] = relu ? fmaxf(y, 0.f) : y;
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
    const long waves = (long)cdiv(M, RM) * cdiv(N, CN);
    hipLaunchKernelGGL(gemm_small_kernel, dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, (hipStream_t)stream, A, a_rows, lda,
                       W, ldw, scale, shift, R, ldr, relu, C, ldc, M, N, K);
    return gom_launch_status();
}
