// The wave- and workgroup-sized pieces of a long-term match (lstmatcher.py:333-381, transformer.py:60-96,
// gom_lstmatcher.py:429-445/510-547) as device functions over a VIRTUAL task index, used by the one-kernel-per-step launches of the
// chain (gemm_small.hip, attn.hip, track.hip: task = the hardware wave / block).  (Round 5 also walked the same tasks from ONE
// persistent launch with grid barriers -- bit-identical, slower, out of the build since round 6: tools/exp/match_fused/.)
// An output's arithmetic is a function of the task alone (its rows, columns and K; never of which wave runs it, or beside what).
#pragma once
#include "common.h"

namespace gom_tasks {

constexpr int RM = 8;                                                  // rows of A per wave
constexpr int CN = 8;                                                  // output columns per wave

__device__ __forceinline__ long gemm_small_tasks(int M, int N) { return (long)((M + RM - 1) / RM) * ((N + CN - 1) / CN); }

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
// Task o = (row group, column group), column group fastest; o < gemm_small_tasks(M, N).
__device__ __forceinline__ void gemm_small_task(const float* __restrict__ A, const int* __restrict__ a_rows, int lda,
                                                const float* __restrict__ W, int ldw, const float* __restrict__ scale,
                                                const float* __restrict__ shift, const float* __restrict__ R, int ldr, int relu,
                                                float* __restrict__ C, int ldc, int M, int N, int K, long o, int lane) {
    const int col_groups = (N + CN - 1) / CN;
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

// Wave-per-query-row attention for the matcher transformers' TINY problems (head_dim 128, at most 64 keys: the long-term match of
// a frame sees 9-53 detections in its window, roi_heads/transformer.py:208,287).  The tile kernel of attn.hip runs such a problem
// as `heads` workgroups with five barriers and three staged phases: 25 us of latency for microseconds of work.  Here one wave
// owns one (batch, head, query row): lane j scores key j (sequential fmaf over the 128 channels, q pre-scaled -- the tile
// kernel's chain), max / sum are 64-lane butterflies, and the output row is accumulated over the keys in ascending order with
// p_j read from lane j as a scalar: the same arithmetic, value for value, as mha_core_kernel, without LDS or barriers.
// Task w = (batch, head, query row), query row fastest; w < batches * heads * Lq.
__device__ __forceinline__ void mha_tiny128_task(const float* __restrict__ q, const float* __restrict__ k,
                                                 const float* __restrict__ v, float* __restrict__ o, int Lq, int Lk, int inner,
                                                 int heads, long q_bo, long q_bi, long q_ss, long k_bo, long k_bi, long k_ss,
                                                 long v_bo, long v_bi, long v_ss, long o_bo, long o_bi, long o_ss, float scale,
                                                 long w, int lane) {
    constexpr int HD = 128;
    const int i = (int)(w % Lq), h = (int)((w / Lq) % heads);
    const long b = w / ((long)Lq * heads), bo = b / inner, bi = b % inner;
    const float* qr = q + bo * q_bo + bi * q_bi + (long)i * q_ss + h * HD;
    const float* kb = k + bo * k_bo + bi * k_bi + h * HD;
    const float* vb = v + bo * v_bo + bi * v_bi + h * HD;
    const int j = lane < Lk ? lane : Lk - 1;                 // idle lanes recompute the last key (discarded)
    const float* kr = kb + (long)j * k_ss;
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < HD; d += 4) {
        const f32x4 qv = *reinterpret_cast<const f32x4*>(qr + d) * scale;
        const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + d);
        s = fmaf(qv[0], kv[0], s);
        s = fmaf(qv[1], kv[1], s);
        s = fmaf(qv[2], kv[2], s);
        s = fmaf(qv[3], kv[3], s);
    }
    if (lane >= Lk) s = -INFINITY;
    const float mx = wave_max(s);
    const float e = lane < Lk ? expf(s - mx) : 0.f;
    const float inv = 1.f / wave_sum(e);
    const float p = e * inv;
    float o0 = 0.f, o1 = 0.f;
    for (int jj = 0; jj < Lk; ++jj) {
        const float pj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), jj));
        const float* vr = vb + (long)jj * v_ss;
        o0 = fmaf(pj, vr[lane], o0);
        o1 = fmaf(pj, vr[lane + 64], o1);
    }
    float* orow = o + bo * o_bo + bi * o_bi + (long)i * o_ss + h * HD;
    orow[lane] = o0;
    orow[lane + 64] = o1;
}

// A match with hoisted projections (matcher_rt.cpp): item i (one 16-byte quad) of the window's embeddings [N, d], their precomputed
// encoder in-projections [N, 3d] and the current frame's precomputed decoder query projections [n_k, d] (rows lo..hi-1 of the
// window); i < (4 N + n_k) * d4.
__device__ __forceinline__ void gather_match_item(const float* __restrict__ pool, int ld_pool, const float* __restrict__ proj,
                                                  int ld_proj, const int* __restrict__ rows, int N, int lo, int n_k, int d4,
                                                  float* __restrict__ src, float* __restrict__ qkv, float* __restrict__ qdec,
                                                  long i) {
    const long n_src = (long)N * d4, n_qkv = (long)N * 3 * d4, n_q = (long)n_k * d4;
    if (i < n_src) {
        const int r = (int)(i / d4), c = (int)(i % d4);
        *reinterpret_cast<f32x4*>(src + i * 4) = *reinterpret_cast<const f32x4*>(pool + (size_t)rows[r] * ld_pool + c * 4);
    } else if (i < n_src + n_qkv) {
        const long k = i - n_src;
        const int r = (int)(k / (3 * d4)), c = (int)(k % (3 * d4));
        *reinterpret_cast<f32x4*>(qkv + k * 4) = *reinterpret_cast<const f32x4*>(proj + (size_t)rows[r] * ld_proj + c * 4);
    } else if (i < n_src + n_qkv + n_q) {
        const long k = i - n_src - n_qkv;
        const int r = (int)(k / d4), c = (int)(k % d4);
        *reinterpret_cast<f32x4*>(qdec + k * 4) =
            *reinterpret_cast<const f32x4*>(proj + (size_t)rows[lo + r] * ld_proj + (3 * d4 + c) * 4);
    }
}

// meta layout (int32): nonk[Np] | col_of[Np] | last_idx[M] | k_inds[n_k]
// Trajectory score of (current detection i, track m) from the activation row of i (global memory or LDS: generic pointer).
// NOT inlined on purpose: every caller (the two-launch form, the one-launch form, the fused chain) runs the same machine code, so
// they agree bit for bit whatever the compiler would hoist, contract or reassociate in either caller.
__device__ __noinline__ static float track_score_one(const float* act_row, const int* __restrict__ meta,
                                                      const float* __restrict__ decay, const float* __restrict__ boxes, float img_w,
                                                      float img_h, int i, int m, int Np, int M, int with_iou,
                                                      float max_center_dist) {
    const int* nonk = meta;
    const int* col_of = meta + Np;
    const int* last_idx = meta + 2 * Np;
    const int* k_inds = meta + 2 * Np + M;
    const float* kb = boxes + (size_t)k_inds[i] * 4;
    const float kx0 = kb[0] / img_w, ky0 = kb[1] / img_h, kx1 = kb[2] / img_w, ky1 = kb[3] / img_h;
    float s = 0.f;
    bool any_valid = false;
    const float kcx = (kx0 + kx1) / 2.f, kcy = (ky0 + ky1) / 2.f;
    const float ks = (kx1 - kx0) * (kx1 - kx0) + (ky1 - ky0) * (ky1 - ky0);
    for (int j = 0; j < Np; ++j) {
        if (col_of[j] != m) continue;
        float a = act_row[nonk[j]];
        if (decay) a *= decay[j];
        s += a;
        if (max_center_dist > 0.f) {
            const float* nb = boxes + (size_t)nonk[j] * 4;
            const float nx0 = nb[0] / img_w, ny0 = nb[1] / img_h, nx1 = nb[2] / img_w, ny1 = nb[3] / img_h;
            const float dx = kcx - (nx0 + nx1) / 2.f, dy = kcy - (ny0 + ny1) / 2.f;
            if ((dx * dx + dy * dy) / (ks + 1e-8f) < max_center_dist) any_valid = true;
        }
    }
    if (with_iou) {
        const float* lb = boxes + (size_t)nonk[last_idx[m]] * 4;
        const float lx0 = lb[0] / img_w, ly0 = lb[1] / img_h, lx1 = lb[2] / img_w, ly1 = lb[3] / img_h;
        const float w = fmaxf(fminf(kx1, lx1) - fmaxf(kx0, lx0), 0.f);
        const float h = fmaxf(fminf(ky1, ly1) - fmaxf(ky0, ly0), 0.f);
        const float inter = w * h;
        const float a1 = (kx1 - kx0) * (ky1 - ky0), a2 = (lx1 - lx0) * (ly1 - ly0);
        const float iou = inter > 0.f ? inter / (a1 + a2 - inter) : 0.f;
        s = fmaxf(s, iou);
    }
    if (max_center_dist > 0.f && !any_valid) s = 0.f;
    return s;
}

// asso_activate + track_score of current detection i by one workgroup of `nthreads` threads (a multiple of 64): the waves run the
// per-frame softmax with the appended zero logit (lstmatcher.py:373-381) over the frame segments into the LDS row `act` [N], then
// the threads call `track_score_one` over the tracks on that row.  A segment's arithmetic is one wave's, a score's one thread's:
// independent of the workgroup's size.  Ends with a barrier (the row may be reused).
__device__ __forceinline__ void asso_score_block(const float* __restrict__ logits, int ld, const int* __restrict__ offs, int T,
                                                 const int* __restrict__ meta, const float* __restrict__ decay,
                                                 const float* __restrict__ boxes, float img_w, float img_h, int Np, int M,
                                                 int with_iou, float max_center_dist, float* __restrict__ traj, int i, float* act,
                                                 int tid, int nthreads) {
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    const float* row = logits + (size_t)i * ld;
    for (int t = wave; t < T; t += nwaves) {
        const int lo = offs[t], hi = offs[t + 1];
        float mx = 0.f;                                      // the appended background logit
        for (int j = lo + lane; j < hi; j += 64) mx = fmaxf(mx, row[j]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lo + lane; j < hi; j += 64) sum += expf(row[j] - mx);
        sum = wave_sum(sum) + expf(0.f - mx);
        for (int j = lo + lane; j < hi; j += 64) act[j] = expf(row[j] - mx) / sum;
    }
    __syncthreads();
    for (int m = tid; m < M; m += nthreads)
        traj[(size_t)i * M + m] = track_score_one(act, meta, decay, boxes, img_w, img_h, i, m, Np, M, with_iou, max_center_dist);
    __syncthreads();
}

}  // namespace gom_tasks
