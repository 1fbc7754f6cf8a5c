// Host-side runtime of one association match: the whole device chain of
// `LSTMatcher._forward_transformer` + `_activate_asso` + the trajectory score
// (lstmatcher.py:333-381, transformer.py:60-96, gom_lstmatcher.py:429-445/510-547) queued from native code with ONE
// call across the FFI.  The tracker is a per-frame serial recurrence of ~18 tiny kernels; issued one by one from
// Python each costs 15-20 us of interpreter + ctypes time, which -- not the GPU -- bounded the replicated tracker of the
// multi-GPU path (0.8 ms/frame).  Nothing here touches the device directly: it only sequences the library's own
// extern "C" entry points on the caller's stream, with the same kernel-selection rule as the Python composition
// (M <= 128 rows -> one-wave-per-column VALU kernel, otherwise the exact-fp32 MFMA GEMM), so both paths return the
// same bits.
#include <stddef.h>

#include "../../include/gomatching_hip.h"

namespace {

struct Lin {
    const float* w;
    const float* b;
};

// Same rule as ops.gemm(..., small=True): <= 64 rows or <= 2^16 outputs -> the patch-per-wave VALU kernel (a long-term match
// of the bench's clips has 9-53 rows: weight streaming, ~5 us per layer), otherwise the deterministic
// split-K exact-fp32 MFMA GEMM up to 1024 rows (tools/skinny_bench.py: 20-50 us against 85-95 us for the whole-K
// loop), the plain exact-fp32 MFMA GEMM beyond.
struct Ctx {
    void* splitk_ws;
    long splitk_bytes;
    void* stream;
};

int linear(const Ctx& c, const float* A, int lda, int M, Lin l, int N, int K, const float* R, int ldr, int relu, float* C,
           int ldc) {
    if (M <= 0) return GOM_OK;
    if (M <= 64 || (long)M * N <= (1L << 16))
        return gom_gemm_small_f32(A, nullptr, lda, l.w, K, nullptr, l.b, R, ldr, relu, C, ldc, M, N, K, c.stream);
    if (M <= 1024)
        return gom_gemm_f32_splitk(A, nullptr, lda, l.w, K, nullptr, l.b, R, ldr, relu, C, ldc, M, N, K, c.splitk_ws,
                                   c.splitk_bytes, c.stream);
    return gom_gemm_f32(A, nullptr, nullptr, lda, l.w, K, nullptr, l.b, R, ldr, relu, C, ldc, M, N, K, c.stream);
}

int attend(const float* q, int ld_q, const float* k, const float* v, int ld_kv, float* o, int d, int heads, int Lq,
           int Lk, void* s) {
    const long st[12] = {0, 0, ld_q, 0, 0, ld_kv, 0, 0, ld_kv, 0, 0, d};
    return gom_mha_core_f32(q, k, v, o, 1, 1, heads, d / heads, Lq, Lk, st, s);
}

#define RT_TRY(expr)                  \
    do {                              \
        const int rc_ = (expr);       \
        if (rc_ != GOM_OK) return rc_; \
    } while (0)

}  // namespace

namespace {
long splitk_floats(int N, int n_k, int d, int ffn) {
    const long wide = 3L * d > ffn ? 3L * d : ffn;
    const int Nc = N < 1024 ? N : 1024, kc = n_k < 1 ? 1 : (n_k < 1024 ? n_k : 1024);   // split-K only runs up to 1024 rows
    long b = gom_gemm_splitk_workspace_bytes(Nc, (int)wide, d);
    const long b2 = ffn > 0 ? gom_gemm_splitk_workspace_bytes(Nc, d, ffn) : 0;
    const long b3 = gom_gemm_splitk_workspace_bytes(kc, N, d);
    b = b > b2 ? b : b2;
    b = b > b3 ? b : b3;
    return (b + 3) / 4;
}
}  // namespace

extern "C" long gom_match_workspace_floats(int N, int n_k, int d, int ffn) {
    if (N <= 0 || n_k < 0 || d <= 0 || ffn < 0) return -1;
    const long wide = 3L * d > ffn ? 3L * d : ffn;
    // src | memory a,b,c | wide scratch (qkv / kv / ffn hidden) | attention out | tgt a,b | q | tgt-side hidden | logits | act
    // | split-K partial sums
    return (long)N * d * 4 + (long)N * wide + (long)N * d + (long)n_k * d * 3 + (long)n_k * wide + 2L * n_k * N + 64 +
           splitk_floats(N, n_k, d, ffn);
}

extern "C" int gom_match_scores_f32(const float* pool, int ld_pool, const int* rows, const int* frame_offsets,
                                    const int* meta, const float* boxes, const float* decay, int N, int T, int lo,
                                    int hi, int num_tracks, const gom_matcher_layer* enc, int n_enc,
                                    const gom_matcher_layer* dec, int n_dec, int d, int heads, int ffn, float img_w,
                                    float img_h, int with_iou, float max_center_dist, float* workspace,
                                    long workspace_floats, float* traj, void* stream) {
    return gom_match_scores_proj_f32(pool, ld_pool, nullptr, 0, rows, frame_offsets, meta, boxes, decay, N, T, lo, hi, num_tracks,
                                     enc, n_enc, dec, n_dec, d, heads, ffn, img_w, img_h, with_iou, max_center_dist, workspace,
                                     workspace_floats, traj, stream);
}

/* The same chain with the two per-row projections of the RAW embeddings hoisted out of it: proj [pool rows, ld_proj >= 4d] holds
 * for every pool row its encoder-layer-0 in-projection (columns [0, 3d): x in_w^T + in_b) and its decoder-layer-0 query
 * projection (columns [3d, 4d)), computed once per detection by the same small-GEMM kernel the chain would use (an output of
 * that kernel depends only on its row, its column and K: the bits are the same).  They are the largest product of a match and
 * one more, i.e. two of its dependent launches -- per frame, 64 times per 8-GPU step.  proj == NULL: everything in the chain. */
extern "C" int gom_match_scores_proj_f32(const float* pool, int ld_pool, const float* proj, int ld_proj, const int* rows,
                                         const int* frame_offsets, const int* meta, const float* boxes, const float* decay,
                                         int N, int T, int lo, int hi, int num_tracks, const gom_matcher_layer* enc,
                                         int n_enc, const gom_matcher_layer* dec, int n_dec, int d, int heads, int ffn,
                                         float img_w, float img_h, int with_iou, float max_center_dist, float* workspace,
                                         long workspace_floats, float* traj, void* stream) {
    if (!pool || !rows || !frame_offsets || !meta || !boxes || !workspace || !traj) return GOM_ERR_INVALID_ARG;
    if (N <= 0 || T <= 0 || lo < 0 || hi <= lo || hi > N || num_tracks <= 0 || d <= 0 || heads <= 0 || d % heads)
        return GOM_ERR_INVALID_ARG;
    if ((n_enc > 0 && !enc) || (n_dec > 0 && !dec) || n_enc < 0 || n_dec < 0) return GOM_ERR_INVALID_ARG;
    const int n_k = hi - lo;
    if (workspace_floats < gom_match_workspace_floats(N, n_k, d, ffn)) return GOM_ERR_INVALID_ARG;
    const long wide = 3L * d > ffn ? 3L * d : ffn;
    float* p = workspace;
    float* src = p;          p += (long)N * d;
    float* mem_a = p;        p += (long)N * d;
    float* mem_b = p;        p += (long)N * d;
    float* mem_c = p;        p += (long)N * d;
    float* big = p;          p += (long)N * wide;
    float* att = p;          p += (long)N * d;
    float* tgt_a = p;        p += (long)n_k * d;
    float* tgt_b = p;        p += (long)n_k * d;
    float* qbuf = p;         p += (long)n_k * d;
    float* hid = p;          p += (long)n_k * wide;
    float* logits = p;       p += (long)n_k * N;
    float* act = p;          p += (long)n_k * N;
    p += (64 - ((p - workspace) & 63)) & 63;                 // 256-byte aligned
    const Ctx ctx{p, 4 * splitk_floats(N, n_k, d, ffn), stream};

    if (proj) {
        if (ld_proj < 4 * d || ld_pool < d) return GOM_ERR_INVALID_ARG;
        RT_TRY(gom_gather_match_f32(pool, ld_pool, proj, ld_proj, rows, N, lo, n_k, d, src, big, qbuf, stream));
    } else {
        if (ld_pool != d) return GOM_ERR_INVALID_ARG;        // gom_gather_rows_f32 reads dense rows
        RT_TRY(gom_gather_rows_f32(pool, rows, src, N, d, stream));
    }
    const float* memory = src;                               // src itself stays intact: the decoder's tgt is a slice of it
    for (int l = 0; l < n_enc; ++l) {                        // post-norm layer with Identity norms (transformer.py:180-195)
        const gom_matcher_layer& L = enc[l];
        if (!L.in_w || !L.out_w || !L.lin1_w || !L.lin2_w) return GOM_ERR_INVALID_ARG;
        float* out = (memory == mem_b) ? mem_c : mem_b;      // never the buffer this layer reads
        if (!(proj && l == 0))                                // layer 0's q | k | v of the raw embeddings: gathered above
            RT_TRY(linear(ctx, memory, d, N, Lin{L.in_w, L.in_b}, 3 * d, d, nullptr, 0, 0, big, 3 * d));
        RT_TRY(attend(big, 3 * d, big + d, big + 2 * d, 3 * d, att, d, heads, N, N, stream));
        RT_TRY(linear(ctx, att, d, N, Lin{L.out_w, L.out_b}, d, d, memory, d, 0, mem_a, d));
        RT_TRY(linear(ctx, mem_a, d, N, Lin{L.lin1_w, L.lin1_b}, ffn, d, nullptr, 0, 1, big, ffn));
        RT_TRY(linear(ctx, big, ffn, N, Lin{L.lin2_w, L.lin2_b}, d, ffn, mem_a, d, 0, out, d));
        memory = out;
    }
    const float* tgt = src + (long)lo * d;                   // tgt = src[query rows] (transformer.py:80-84)
    for (int l = 0; l < n_dec; ++l) {                        // cross-attention only (transformer.py:270-294)
        const gom_matcher_layer& L = dec[l];
        if (!L.in_w || !L.out_w) return GOM_ERR_INVALID_ARG;
        if (!(proj && l == 0))                                // layer 0's query projection of the raw embeddings: gathered above
            RT_TRY(linear(ctx, tgt, d, n_k, Lin{L.in_w, L.in_b}, d, d, nullptr, 0, 0, qbuf, d));
        RT_TRY(linear(ctx, memory, d, N, Lin{L.in_w + (size_t)d * d, L.in_b ? L.in_b + d : nullptr}, 2 * d, d, nullptr, 0, 0,
                      big, 2 * d));
        RT_TRY(attend(qbuf, d, big, big + d, 2 * d, att, d, heads, n_k, N, stream));
        float* out = (tgt == tgt_a) ? tgt_b : tgt_a;
        RT_TRY(linear(ctx, att, d, n_k, Lin{L.out_w, L.out_b}, d, d, tgt, d, 0, out, d));
        tgt = out;
        if (L.lin1_w) {
            if (!L.lin2_w) return GOM_ERR_INVALID_ARG;
            float* out2 = (tgt == tgt_a) ? tgt_b : tgt_a;
            RT_TRY(linear(ctx, tgt, d, n_k, Lin{L.lin1_w, L.lin1_b}, ffn, d, nullptr, 0, 1, hid, ffn));
            RT_TRY(linear(ctx, hid, ffn, n_k, Lin{L.lin2_w, L.lin2_b}, d, ffn, tgt, d, 0, out2, d));
            tgt = out2;
        }
    }
    // ATTWeightHead with 0 layers: q . k^T (lstmatcher.py:360-371)
    RT_TRY(linear(ctx, tgt, d, n_k, Lin{memory, nullptr}, N, d, nullptr, 0, 0, logits, N));
    if (N <= 16384)                                          // activation + trajectory score as one launch (same values)
        return gom_asso_score_f32(logits, N, frame_offsets, T, meta, decay, boxes, img_w, img_h, n_k, N - n_k, num_tracks,
                                  with_iou, max_center_dist, traj, stream);
    RT_TRY(gom_asso_activate_f32(logits, N, frame_offsets, T, n_k, act, N, stream));
    return gom_track_score_f32(act, N, meta, decay, boxes, img_w, img_h, n_k, N - n_k, num_tracks, with_iou,
                               max_center_dist, traj, stream);
}
