// Tiny products of the tracker path: association logits  C[n_cur, N] = tgt . memory^T  (transformer.py:92-96 /
// lstmatcher.py:360-371) and similar GEMMs with at most a few thousand outputs and K up to a few thousand.  A 256x64
// MFMA tile spends 16 blocks x K/2 exact-fp32 MFMAs on such a product whatever M is (90 us at M=5, N=55, K=1024);
// and the skinny M <= 128 linear layers of the matcher transformers pay a two-kernel split-K (22 us) each.  Here one
// wave owns one output column for 8 rows on the VALU: latency of a few us, deterministic, independent of M.
#include "common.h"
#include "tracker_tasks.h"

namespace {

using gom_tasks::CN;
using gom_tasks::RM;

// (the patch-per-wave arithmetic lives in tracker_tasks.h: the fused match kernel runs the same tasks)
__global__ __launch_bounds__(256) void gemm_small_kernel(const float* __restrict__ A, const int* __restrict__ a_rows,
                                                         int lda, const float* __restrict__ W, int ldw,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ R, int ldr, int relu,
                                                         float* __restrict__ C, int ldc, int M, int N, int K) {
    const int lane = threadIdx.x & 63;
    const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // (row group, column group), column group fastest
    if (o >= gom_tasks::gemm_small_tasks(M, N)) return;
    gom_tasks::gemm_small_task(A, a_rows, lda, W, ldw, scale, shift, R, ldr, relu, C, ldc, M, N, K, o, lane);
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
