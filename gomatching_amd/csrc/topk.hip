// Top-k proposal selection over the S encoder tokens (deformable_transformer.py:188-190):
//   topk_proposals = torch.topk(enc_outputs_class[..., 0], nq, dim=1)[1]
// Two LDS bitonic-sort stages: per-chunk top-k, then a merge of the chunk winners.  Keys are
// (order-preserving float bits, ~index) packed in 64 bits, so the result is sorted by value
// descending with ties resolved towards the LOWER token index (deterministic; the reference's tie
// order is unspecified).  Invalid-proposal tokens get the constant logit the reference produces for
// a zeroed memory row (see DESIGN.md "proposal masking").
#include "common.h"

namespace {

constexpr int CHUNK = 4096;
constexpr int MERGE_MAX = 8192;

__device__ __forceinline__ unsigned long long make_key(float v, unsigned idx) {
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);        // ascending-orderable
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}

template <int N>
__device__ void bitonic_sort_desc(unsigned long long* keys) {   // N power of two, blockDim = 1024
    for (int size = 2; size <= N; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < N / 2; t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void topk_chunk_kernel(const float* __restrict__ logits, int ld,
                                                          const unsigned char* __restrict__ valid,
                                                          const float* __restrict__ invalid_logit, long S, int k,
                                                          unsigned long long* __restrict__ cand, int chunks) {
    __shared__ unsigned long long keys[CHUNK];
    const int b = blockIdx.y, ch = blockIdx.x;
    const long s0 = (long)ch * CHUNK;
    const float c0 = invalid_logit ? invalid_logit[0] : 0.f;
    for (int t = threadIdx.x; t < CHUNK; t += blockDim.x) {
        const long s = s0 + t;
        unsigned long long key = 0ull;                       // below every real key
        if (s < S) {
            float v = logits[((size_t)b * S + s) * ld];
            if (valid && !valid[s]) v = c0;
            key = make_key(v, (unsigned)s);
        }
        keys[t] = key;
    }
    bitonic_sort_desc<CHUNK>(keys);
    unsigned long long* out = cand + ((size_t)b * chunks + ch) * k;
    for (int t = threadIdx.x; t < k; t += blockDim.x) out[t] = keys[t];
}

__global__ __launch_bounds__(1024) void topk_merge_kernel(const unsigned long long* __restrict__ cand, int ncand, int k,
                                                          long S, int* __restrict__ idx_out,
                                                          int* __restrict__ rows_out) {
    extern __shared__ unsigned long long mkeys[];
    const int b = blockIdx.x;
    for (int t = threadIdx.x; t < MERGE_MAX; t += blockDim.x) mkeys[t] = t < ncand ? cand[(size_t)b * ncand + t] : 0ull;
    bitonic_sort_desc<MERGE_MAX>(mkeys);
    for (int t = threadIdx.x; t < k; t += blockDim.x) {
        const int s = (int)(0xFFFFFFFFu - (unsigned)(mkeys[t] & 0xFFFFFFFFull));
        idx_out[(size_t)b * k + t] = s;
        if (rows_out) rows_out[(size_t)b * k + t] = (int)(b * S + s);   // row into the [B*S, .] token buffers
    }
}

}  // namespace

extern "C" long gom_topk_workspace_bytes(int B, long S, int k) {
    return (long)sizeof(unsigned long long) * B * cdiv(S, CHUNK) * k;
}

extern "C" int gom_topk_tokens(const float* logits, int ld, const unsigned char* valid, const float* invalid_logit,
                               int B, long S, int k, void* workspace, int* idx_out, int* rows_out, void* stream) {
    GOM_CHECK_ARG(logits && workspace && idx_out && B > 0 && S > 0 && k > 0 && ld >= 1);
    const int chunks = cdiv(S, CHUNK);
    GOM_CHECK_ARG(k <= CHUNK && k <= S && (long)chunks * k <= MERGE_MAX && (long)B * S < (1L << 31));
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* cand = (unsigned long long*)workspace;
    hipLaunchKernelGGL(topk_chunk_kernel, dim3((unsigned)chunks, (unsigned)B), dim3(1024), 0, s, logits, ld, valid,
                       invalid_logit, S, k, cand, chunks);
    auto kern = topk_merge_kernel;
    const int lds = MERGE_MAX * sizeof(unsigned long long);
    // (the attribute is per DEVICE: set on every launch -- a process-wide flag would miss a second GPU; it costs ~1 us)
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(1024), lds, s, cand, chunks * k, k, S, idx_out, rows_out);
    return gom_launch_status();
}
