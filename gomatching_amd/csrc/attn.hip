// Softmax attention core for the short sequences of the GoMatching path (fp32, exact softmax):
//   * DeepSolo decoder intra-instance attention  (25 points  x batch nq,  8 heads x 32)
//   * DeepSolo decoder inter-instance attention  (nq queries x batch 25,  8 heads x 32)
//     (deformable_transformer.py:386-404, nn.MultiheadAttention core)
//   * matcher transformers (N <= 6*nq detections, 8 heads x 128)   (roi_heads/transformer.py:208,287)
// Projections are done by gom_gemm_f32; this kernel is  O = softmax(Q K^T) V  per (batch, head) with
// generic element strides so the seq-first / batch-swapped views of the reference need no copies.
// One workgroup = QT query rows x all keys; the score rows live in LDS (no HBM round trip).
#include "common.h"
#include "tracker_tasks.h"

namespace {

constexpr int KT = 64;  // keys per staged tile

template <int HD, int QT>
__global__ __launch_bounds__(256) void mha_core_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, float* __restrict__ o,
                                                       int Lq, int Lk, int inner, long q_bo, long q_bi, long q_ss,
                                                       long k_bo, long k_bi, long k_ss, long v_bo, long v_bi,
                                                       long v_ss, long o_bo, long o_bi, long o_ss, float scale,
                                                       const int* __restrict__ seg) {
    constexpr int KS = HD + 4;  // padded K-tile row (conflict-free ds_read_b128 across keys)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // ragged batch: block x = segment s with its own (first query row, Lq, first key row, Lk); q_bo / k_bo / v_bo / o_bo
    // are then the ROW strides applied to those first rows
    if (seg) {
        const int* d = seg + 4 * blockIdx.x;
        Lq = d[1];
        Lk = d[3];
        if ((int)blockIdx.z * QT >= Lq || Lk <= 0) return;      // uniform per block: no barrier is skipped by a subset
    }
    const int Lkp = (Lk + KT - 1) / KT * KT;
    float* Qs = smem;                       // [QT][HD]
    float* KVs = Qs + QT * HD;              // [KT][KS]
    float* Ss = KVs + KT * KS;              // [QT][Lkp]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.z * QT, h = blockIdx.y;
    const long bo = blockIdx.x / inner, bi = blockIdx.x % inner;   // batch = outer x inner, two strides each
    const float* qb = q + (seg ? seg[4 * blockIdx.x] * q_ss : bo * q_bo + bi * q_bi) + h * HD;
    const float* kb = k + (seg ? seg[4 * blockIdx.x + 2] * k_ss : bo * k_bo + bi * k_bi) + h * HD;
    const float* vb = v + (seg ? seg[4 * blockIdx.x + 2] * v_ss : bo * v_bo + bi * v_bi) + h * HD;
    float* ob = o + (seg ? seg[4 * blockIdx.x] * o_ss : bo * o_bo + bi * o_bi) + h * HD;

    for (int u = tid; u < QT * HD / 4; u += 256) {
        const int r = u / (HD / 4), d4 = (u % (HD / 4)) * 4;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (i0 + r < Lq) x = *reinterpret_cast<const f32x4*>(qb + (i0 + r) * q_ss + d4) * scale;
        *reinterpret_cast<f32x4*>(Qs + r * HD + d4) = x;
    }

    // ---- S = (scale*Q) K^T -------------------------------------------------------------------
    constexpr int RPT = QT / 4;  // score rows per thread (rows wave, wave+4, ...)
    for (int kt = 0; kt < Lkp; kt += KT) {
        __syncthreads();
        for (int u = tid; u < KT * HD / 4; u += 256) {
            const int r = u / (HD / 4), d4 = (u % (HD / 4)) * 4;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (kt + r < Lk) x = *reinterpret_cast<const f32x4*>(kb + (kt + r) * k_ss + d4);
            *reinterpret_cast<f32x4*>(KVs + r * KS + d4) = x;
        }
        __syncthreads();
        float acc[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) acc[r] = 0.f;
#pragma unroll 4
        for (int d = 0; d < HD; d += 4) {
            const f32x4 kv = *reinterpret_cast<const f32x4*>(KVs + lane * KS + d);
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const f32x4 qv = *reinterpret_cast<const f32x4*>(Qs + (wave + 4 * r) * HD + d);
                acc[r] = fmaf(qv[0], kv[0], acc[r]);
                acc[r] = fmaf(qv[1], kv[1], acc[r]);
                acc[r] = fmaf(qv[2], kv[2], acc[r]);
                acc[r] = fmaf(qv[3], kv[3], acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) Ss[(wave + 4 * r) * Lkp + kt + lane] = (kt + lane < Lk) ? acc[r] : -INFINITY;
    }
    __syncthreads();

    // ---- row softmax (exact: exp(x - max) / sum) ----------------------------------------------
    for (int r = wave; r < QT; r += 4) {
        float* row = Ss + r * Lkp;
        float mx = -INFINITY;
        for (int j = lane; j < Lk; j += 64) mx = fmaxf(mx, row[j]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < Lk; j += 64) {
            const float e = expf(row[j] - mx);
            row[j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int j = lane; j < Lkp; j += 64) row[j] = (j < Lk) ? row[j] * inv : 0.f;
    }

    // ---- O = P V ---------------------------------------------------------------------------------
    constexpr int ROWS_PER_PASS = 256 / HD;          // distinct rows covered by the block at once
    constexpr int OPT = QT / ROWS_PER_PASS;          // outputs per thread
    const int d = tid % HD, rg = tid / HD;
    float oacc[OPT];
#pragma unroll
    for (int r = 0; r < OPT; ++r) oacc[r] = 0.f;
    for (int kt = 0; kt < Lkp; kt += KT) {
        __syncthreads();
        for (int u = tid; u < KT * HD / 4; u += 256) {
            const int r = u / (HD / 4), d4 = (u % (HD / 4)) * 4;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (kt + r < Lk) x = *reinterpret_cast<const f32x4*>(vb + (kt + r) * v_ss + d4);
            *reinterpret_cast<f32x4*>(KVs + r * KS + d4) = x;
        }
        __syncthreads();
#pragma unroll 8
        for (int j = 0; j < KT; ++j) {
            const float vv = KVs[j * KS + d];
#pragma unroll
            for (int r = 0; r < OPT; ++r) oacc[r] = fmaf(Ss[(rg + r * ROWS_PER_PASS) * Lkp + kt + j], vv, oacc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < OPT; ++r) {
        const int i = i0 + rg + r * ROWS_PER_PASS;
        if (i < Lq) ob[i * o_ss + d] = oacc[r];
    }
}

// Row-per-lane form for the DeepSolo decoder's two attentions (head_dim 32; 25 points x batch B*nq and nq queries x
// batch 25*B, deformable_transformer.py:386-404): the tile kernel above spends a 256-thread workgroup with 64-key
// tiles and five barriers on a 25 x 25 problem.  Here one workgroup = one (batch, head), one LANE = one query row with
// q and the output row in registers; K and V of the (batch, head) sit in LDS and are read as wave-uniform (broadcast)
// ds_read_b128.  Two passes over the keys (max, then exp / sum / PV with the scores recomputed) keep the softmax exact
// without storing a score row; normalisation by the row sum is applied to the accumulated output.
template <int HD>
__global__ __launch_bounds__(512) void mha_rows_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, float* __restrict__ o, int Lq, int Lk,
                                                       int inner, long q_bo, long q_bi, long q_ss, long k_bo, long k_bi,
                                                       long k_ss, long v_bo, long v_bi, long v_ss, long o_bo, long o_bi,
                                                       long o_ss, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                       // [Lk][HD]
    float* Vs = smem + (size_t)Lk * HD;     // [Lk][HD]
    const int tid = threadIdx.x, h = blockIdx.y;
    const long bo = blockIdx.x / inner, bi = blockIdx.x % inner;
    const float* qb = q + bo * q_bo + bi * q_bi + h * HD;
    const float* kb = k + bo * k_bo + bi * k_bi + h * HD;
    const float* vb = v + bo * v_bo + bi * v_bi + h * HD;
    float* ob = o + bo * o_bo + bi * o_bi + h * HD;
    for (int u = tid; u < Lk * (HD / 4); u += blockDim.x) {
        const int r = u / (HD / 4), d4 = (u % (HD / 4)) * 4;
        *reinterpret_cast<f32x4*>(Ks + r * HD + d4) = *reinterpret_cast<const f32x4*>(kb + r * k_ss + d4);
        *reinterpret_cast<f32x4*>(Vs + r * HD + d4) = *reinterpret_cast<const f32x4*>(vb + r * v_ss + d4);
    }
    const int i = tid;
    const bool live = i < Lq;
    f32x4 qv[HD / 4];
#pragma unroll
    for (int d = 0; d < HD / 4; ++d)
        qv[d] = live ? *reinterpret_cast<const f32x4*>(qb + (long)i * q_ss + 4 * d) * scale : f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    auto score = [&](int j) {
        const float* kr = Ks + j * HD;
        float s0 = 0.f, s1 = 0.f;                            // two chains: half the dependent-FMA latency
#pragma unroll
        for (int d = 0; d < HD / 4; d += 2) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(kr + 4 * d);
            const f32x4 b = *reinterpret_cast<const f32x4*>(kr + 4 * d + 4);
            s0 = fmaf(qv[d][0], a[0], s0); s0 = fmaf(qv[d][1], a[1], s0);
            s0 = fmaf(qv[d][2], a[2], s0); s0 = fmaf(qv[d][3], a[3], s0);
            s1 = fmaf(qv[d + 1][0], b[0], s1); s1 = fmaf(qv[d + 1][1], b[1], s1);
            s1 = fmaf(qv[d + 1][2], b[2], s1); s1 = fmaf(qv[d + 1][3], b[3], s1);
        }
        return s0 + s1;
    };
    float mx = -INFINITY;
    for (int j = 0; j < Lk; ++j) mx = fmaxf(mx, score(j));
    f32x4 acc[HD / 4];
#pragma unroll
    for (int d = 0; d < HD / 4; ++d) acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float sum = 0.f;
    for (int j = 0; j < Lk; ++j) {
        const float e = expf(score(j) - mx);
        sum += e;
        const float* vr = Vs + j * HD;
#pragma unroll
        for (int d = 0; d < HD / 4; ++d) acc[d] += *reinterpret_cast<const f32x4*>(vr + 4 * d) * e;
    }
    if (live) {
        const float inv = 1.f / sum;
#pragma unroll
        for (int d = 0; d < HD / 4; ++d) *reinterpret_cast<f32x4*>(ob + (long)i * o_ss + 4 * d) = acc[d] * inv;
    }
}

// Wave-per-query-row form for the matcher transformers' TINY problems (head_dim 128, at most 64 keys): one wave owns one
// (batch, head, query row) -- gom_tasks::mha_tiny128_task (tracker_tasks.h; the fused match kernel runs the same tasks).
__global__ __launch_bounds__(256) void mha_tiny128_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, float* __restrict__ o, int Lq, int Lk,
                                                          int inner, int heads, long q_bo, long q_bi, long q_ss, long k_bo,
                                                          long k_bi, long k_ss, long v_bo, long v_bi, long v_ss, long o_bo,
                                                          long o_bi, long o_ss, float scale, long total) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= total) return;
    gom_tasks::mha_tiny128_task(q, k, v, o, Lq, Lk, inner, heads, q_bo, q_bi, q_ss, k_bo, k_bi, k_ss, v_bo, v_bi, v_ss, o_bo, o_bi,
                                o_ss, scale, w, lane);
}

template <int HD, int QT>
int launch(const float* q, const float* k, const float* v, float* o, int outer, int inner, int heads, int Lq, int Lk,
           const long* st, float scale, hipStream_t s, bool* fits, const int* seg = nullptr) {
    const int Lkp = (Lk + KT - 1) / KT * KT;
    const size_t lds = sizeof(float) * ((size_t)QT * HD + (size_t)KT * (HD + 4) + (size_t)QT * Lkp);
    *fits = lds <= 160 * 1024;
    if (!*fits) return GOM_OK;
    auto kern = mha_core_kernel<HD, QT>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(outer * inner), (unsigned)heads, (unsigned)cdiv(Lq, QT)), dim3(256), lds, s,
                       q, k, v, o, Lq, Lk, inner, st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7], st[8], st[9],
                       st[10], st[11], scale, seg);
    return gom_launch_status();
}

}  // namespace

extern "C" int gom_mha_core_f32(const float* q, const float* k, const float* v, float* o, int batch_outer,
                                int batch_inner, int heads, int head_dim, int Lq, int Lk, const long* strides,
                                void* stream) {
    GOM_CHECK_ARG(q && k && v && o && strides);
    GOM_CHECK_ARG(batch_outer >= 0 && batch_inner > 0 && heads > 0 && Lq >= 0 && Lk >= 0);
    GOM_CHECK_ARG(head_dim == 32 || head_dim == 128);
    for (int i = 0; i < 9; ++i) GOM_CHECK_ARG((strides[i] % 4) == 0);        // q, k, v rows are read as float4
    if (batch_outer == 0 || Lq == 0) return GOM_OK;
    GOM_CHECK_ARG(Lk > 0);
    const float scale = 1.0f / sqrtf((float)head_dim);
    hipStream_t s = (hipStream_t)stream;
    bool fits = false;
    int rc;
#define TRY(HD, QT)                                                                                             \
    rc = launch<HD, QT>(q, k, v, o, batch_outer, batch_inner, heads, Lq, Lk, strides, scale, s, &fits);        \
    if (rc != GOM_OK || fits) return rc;
    if (head_dim == 32) {
        const size_t rows_lds = (size_t)2 * Lk * 32 * sizeof(float);
        if (Lq <= 512 && rows_lds <= 128 * 1024 && ((strides[9] | strides[10] | strides[11]) % 4) == 0) {   // lane = query row
            auto kern = mha_rows_kernel<32>;
            if (rows_lds > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)rows_lds);
                if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
            }
            const int threads = (Lq + 63) / 64 * 64;
            hipLaunchKernelGGL(kern, dim3((unsigned)(batch_outer * batch_inner), (unsigned)heads), dim3(threads), rows_lds, s,
                               q, k, v, o, Lq, Lk, batch_inner, strides[0], strides[1], strides[2], strides[3], strides[4],
                               strides[5], strides[6], strides[7], strides[8], strides[9], strides[10], strides[11], scale);
            return gom_launch_status();
        }
        TRY(32, 32) TRY(32, 8)
    } else {
        const long waves = (long)batch_outer * batch_inner * heads * Lq;
        if (Lk <= 64 && waves <= (1L << 16)) {               // tiny: one wave per (batch, head, query row), same arithmetic
            hipLaunchKernelGGL(mha_tiny128_kernel, dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, s, q, k, v, o, Lq, Lk,
                               batch_inner, heads, strides[0], strides[1], strides[2], strides[3], strides[4], strides[5],
                               strides[6], strides[7], strides[8], strides[9], strides[10], strides[11], scale, waves);
            return gom_launch_status();
        }
        TRY(128, 32) TRY(128, 8)
    }
#undef TRY
    return GOM_ERR_UNSUPPORTED;  // more keys than one workgroup's LDS can hold
}

// Ragged batch of independent attention problems over row ranges of the same q / k / v / o matrices (the short-term
// matcher's per-pair attention: transformer.py:60-96 applied to every (previous, current) frame pair of a clip at once).
// segments [device] int32 [S][4] = (first query row, Lq, first key row, Lk); ld_* are row strides in floats; max_Lq /
// max_Lk bound the grid and the LDS score rows.
extern "C" int gom_mha_core_segments_f32(const float* q, const float* k, const float* v, float* o, const int* segments,
                                         int num_segments, int heads, int head_dim, int ld_q, int ld_k, int ld_v,
                                         int ld_o, int max_Lq, int max_Lk, void* stream) {
    GOM_CHECK_ARG(q && k && v && o && segments && num_segments >= 0 && heads > 0 && head_dim == 128);
    GOM_CHECK_ARG((ld_q % 4) == 0 && (ld_k % 4) == 0 && (ld_v % 4) == 0 && max_Lq >= 0 && max_Lk >= 0);
    if (num_segments == 0 || max_Lq == 0 || max_Lk == 0) return GOM_OK;
    const float scale = 1.0f / sqrtf((float)head_dim);
    const long st[12] = {0, 0, ld_q, 0, 0, ld_k, 0, 0, ld_v, 0, 0, ld_o};
    bool fits = false;
    int rc = launch<128, 32>(q, k, v, o, num_segments, 1, heads, max_Lq, max_Lk, st, scale, (hipStream_t)stream, &fits,
                             segments);
    if (rc != GOM_OK || fits) return rc;
    rc = launch<128, 8>(q, k, v, o, num_segments, 1, heads, max_Lq, max_Lk, st, scale, (hipStream_t)stream, &fits, segments);
    if (rc != GOM_OK || fits) return rc;
    return GOM_ERR_UNSUPPORTED;
}
