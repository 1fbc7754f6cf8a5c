// Multi-scale deformable attention sampling (forward) for gfx950.
//
// Drop-in for the reference's only native op, `adet._C.ms_deform_attn_forward`
// (/root/reference/third_party/adet/layers/csrc/DeformAttn/ms_deform_attn.h:20-39, kernel
// ms_deform_im2col_cuda.cuh:237-299, bilinear :33-84):
//     out[b,q,m,:] = sum_{l,p} w[b,q,m,l,p] * bilinear(value_l[b,:,m,:], loc[b,q,m,l,p])
// with zero padding outside the level map and the kernel's (-1,H)x(-1,W) acceptance window.
//
// The reference maps one thread to one output ELEMENT (block 1024), so the 32 threads of a head
// re-read the same loc/weight and gather 4-byte corners.  Here one wavefront owns one query: 8
// lanes per head, each lane 4 channels, so every corner fetch is a 16-byte load, a head's corner
// is one 128-byte line, and the query's 256 outputs leave as one contiguous 1 KiB store.  The op
// is gather-bound (HBM/L2), not MFMA work.
#include <mutex>
#include <type_traits>
#include <vector>

#include "common.h"

namespace {

constexpr int HEADS = 8, CH = 32, LEVELS = 4;

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one, MI355X_MICROARCH.md), so with the plain
// blockIdx -> query map every XCD's private L2 sees every 8th group of four queries: spatial neighbours, whose bilinear
// corners are the same cache lines, sit on eight different L2s and every value line is fetched up to eight times.  This
// bijective remap hands each XCD one contiguous eighth of the queries (a band of image rows per level) instead.  Speed only.
__device__ __forceinline__ long xcd_contiguous_block(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (long)(xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int POINTS>
__global__ __launch_bounds__(256) void msda_fwd_kernel(const float* __restrict__ value,
                                                       const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ lsi,
                                                       const float* __restrict__ loc,
                                                       const float* __restrict__ attw, float* __restrict__ out,
                                                       int B, int Lq, long v_bs, int v_rs) {
    const long q_global = xcd_contiguous_block(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);   // one wave per (b, q)
    if (q_global >= (long)B * Lq) return;
    const int lane = threadIdx.x & 63;
    const int m = lane >> 3;            // head
    const int c4 = (lane & 7) * 4;      // first of this lane's 4 channels
    const int b = (int)(q_global / Lq);

    const float* vb = value + (size_t)b * v_bs + m * CH + c4;
    const float* lp = loc + ((size_t)q_global * HEADS + m) * (LEVELS * POINTS * 2);
    const float* wp = attw + ((size_t)q_global * HEADS + m) * (LEVELS * POINTS);

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < LEVELS; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const float* vl = vb + (size_t)lsi[l] * v_rs;
#pragma unroll
        for (int p = 0; p < POINTS; ++p) {
            const float lx = lp[(l * POINTS + p) * 2], ly = lp[(l * POINTS + p) * 2 + 1];
            const float w = wp[l * POINTS + p];
            const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
            if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
                const float lh = h_im - h_low, lw = w_im - w_low;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const bool y0 = h_low >= 0, y1 = h_low + 1 <= H - 1;
                const bool x0 = w_low >= 0, x1 = w_low + 1 <= W - 1;
                const float* base = vl + ((long)h_low * W + w_low) * (long)v_rs;
                f32x4 v1 = {0.f, 0.f, 0.f, 0.f}, v2 = v1, v3 = v1, v4 = v1;
                if (y0 && x0) v1 = *reinterpret_cast<const f32x4*>(base);
                if (y0 && x1) v2 = *reinterpret_cast<const f32x4*>(base + v_rs);
                if (y1 && x0) v3 = *reinterpret_cast<const f32x4*>(base + (long)W * v_rs);
                if (y1 && x1) v4 = *reinterpret_cast<const f32x4*>(base + (long)(W + 1) * v_rs);
                const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                const f32x4 val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
                acc += val * w;
            }
        }
    }
    *reinterpret_cast<f32x4*>(out + (size_t)q_global * (HEADS * CH) + m * CH + c4) = acc;
}

// sampling-location arithmetic + softmax of ms_deform_attn.py:136-145, fused in one pass:
//   raw [Q, 8*(L*P*2) offsets | 8*(L*P) logits] (one GEMM output row) , ref [Q, L, 2]
//   -> loc [Q,8,L,P,2], w [Q,8,L,P]
template <int POINTS>
__global__ __launch_bounds__(256) void msda_prep_kernel(const float* __restrict__ raw, int ld_raw,
                                                        const float* __restrict__ ref, int ref_levels,
                                                        const int64_t* __restrict__ shapes,
                                                        float* __restrict__ loc, float* __restrict__ attw, long Q) {
    constexpr int LP = LEVELS * POINTS;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (q, head)
    if (idx >= Q * HEADS) return;
    const long q = idx / HEADS;
    const int m = (int)(idx % HEADS);
    const float* off = raw + q * ld_raw + m * (LP * 2);
    const float* lg = raw + q * ld_raw + HEADS * LP * 2 + m * LP;
    float e[LP];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < LP; ++i) { e[i] = lg[i]; mx = fmaxf(mx, e[i]); }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LP; ++i) { e[i] = expf(e[i] - mx); sum += e[i]; }
    float* wo = attw + idx * LP;
    float* lo = loc + idx * (LP * 2);
#pragma unroll
    for (int l = 0; l < LEVELS; ++l) {
        const float Hf = (float)shapes[2 * l], Wf = (float)shapes[2 * l + 1];
        const float* r = ref + (q * ref_levels + (ref_levels == 1 ? 0 : l)) * 2;
        const float rx = r[0], ry = r[1];
#pragma unroll
        for (int p = 0; p < POINTS; ++p) {
            const int i = l * POINTS + p;
            wo[i] = e[i] / sum;
            lo[2 * i] = rx + off[2 * i] / Wf;
            lo[2 * i + 1] = ry + off[2 * i + 1] / Hf;
        }
    }
}

// Fused MSDeformAttn core (ms_deform_attn.py:136-151 minus the two projections): softmax over the 16 logits,
// sampling-location arithmetic and the bilinear gather in ONE pass over the [Q, 384] offsets|logits rows, so
// the [Q,8,4,4,2] locations and [Q,8,4,4] weights never exist in HBM (saves 3 x 457 MB of traffic per
// encoder call at 8 x 37 171 tokens and one launch).  Same lane mapping as msda_fwd_kernel.
// acc += w * v, one fused multiply-add per channel.  The fused kernels below accumulate a sample as FOUR of these with the corner
// weights already multiplied by the sample's attention weight (round 6: at the sample's owner, once) -- 16 vector instructions per
// sample and lane where (w1 v1 + w2 v2 + w3 v3 + w4 v4) * w (ms_deform_im2col's order, kept by msda_fwd_kernel) takes 20 and one
// more broadcast.  The same sequence in every fused kernel: they stay bit-identical to one another.
__device__ __forceinline__ void msda_fma4(f32x4& acc, const float w, const f32x4 v) {
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = fmaf(w, v[c], acc[c]);
}

template <int POINTS, bool HAS_VR>
__global__ __launch_bounds__(256) void msda_fused_kernel(const float* __restrict__ value,
                                                         const int64_t* __restrict__ shapes,
                                                         const int64_t* __restrict__ lsi,
                                                         const float* __restrict__ raw, int ld_raw,
                                                         const float* __restrict__ ref, float* __restrict__ out,
                                                         int B, int Lq, long v_bs, int v_rs,
                                                         const float* __restrict__ vr) {
    constexpr int LP = LEVELS * POINTS;
    const long q_global = xcd_contiguous_block(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    if (q_global >= (long)B * Lq) return;
    const int lane = threadIdx.x & 63;
    const int m = lane >> 3;
    const int c4 = (lane & 7) * 4;
    const int b = (int)(q_global / Lq);

    const float* vb = value + (size_t)b * v_bs + m * CH + c4;
    const float* op = raw + (size_t)q_global * ld_raw + m * (LP * 2);
    const float* lp = raw + (size_t)q_global * ld_raw + HEADS * LP * 2 + m * LP;
    f32x4 offv[LP / 2], lgv[LP / 4];
#pragma unroll
    for (int i = 0; i < LP / 2; ++i) offv[i] = *reinterpret_cast<const f32x4*>(op + 4 * i);
#pragma unroll
    for (int i = 0; i < LP / 4; ++i) lgv[i] = *reinterpret_cast<const f32x4*>(lp + 4 * i);
    const float rx = ref[q_global * 2], ry = ref[q_global * 2 + 1];
    float e[LP];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < LP; ++i) { e[i] = lgv[i >> 2][i & 3]; mx = fmaxf(mx, e[i]); }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LP; ++i) { e[i] = expf(e[i] - mx); sum += e[i]; }

    // The loop below is VALU-bound (47 vector instructions per corner load before this form), so:
    //   * off / W and off / H (ms_deform_attn.py:141-147) are divisions by per-level constants: q = x * (1/W) followed by
    //     one residual correction, r = fma(-q, W, x), q += r * (1/W) -- correctly rounded save for rare double-rounding
    //     cases, 3 instructions instead of the ~10 of a full IEEE division;
    //   * the softmax denominator is inverted once (e * (1/sum) differs from e / sum by at most one ulp);
    //   * out-of-range corners are not branched around: the corner index is clamped into the map and its weight zeroed
    //     (0 * finite = 0), which is what the reference's zero padding computes;
    //   * corner offsets are 32-bit element offsets from the level's base (a level holds < 2^31 floats).
    const float inv_sum = 1.f / sum;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < LEVELS; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const float Hf = (float)H, Wf = (float)W;
        const float rW = 1.f / Wf, rH = 1.f / Hf;
        const float* vl = vb + (size_t)lsi[l] * v_rs;
#pragma unroll
        for (int p = 0; p < POINTS; ++p) {
            const int i = l * POINTS + p;
            const float ox = offv[i >> 1][(i & 1) * 2], oy = offv[i >> 1][(i & 1) * 2 + 1];
            float qx = ox * rW, qy = oy * rH;
            qx = fmaf(fmaf(-qx, Wf, ox), rW, qx);
            qy = fmaf(fmaf(-qy, Hf, oy), rH, qy);
            // padded batches: the reference point is scaled by the level's valid ratio first (deformable_transformer.py:
            // 262-263 / 470-472); HAS_VR = false is the unpadded case (ratios 1): a
            // compile-time switch, the run-time form of it cost the unpadded kernel 80 % (503 -> 919 us)
            const float lx = (HAS_VR ? rx * vr[2 * l] : rx) + qx, ly = (HAS_VR ? ry * vr[2 * l + 1] : ry) + qy;
            const float w = e[i] * inv_sum;
            const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
            const bool inside = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const float lh = h_im - hf, lw = w_im - wf;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const int h_low = inside ? (int)hf : 0, w_low = inside ? (int)wf : 0;
            const bool y0 = h_low >= 0, y1 = h_low + 1 <= H - 1;
            const bool x0 = w_low >= 0, x1 = w_low + 1 <= W - 1;
            const int yc0 = y0 ? h_low : 0, yc1 = y1 ? h_low + 1 : H - 1;
            const int xc0 = x0 ? w_low : 0, xc1 = x1 ? w_low + 1 : W - 1;
            const int r0 = yc0 * W, r1 = yc1 * W;
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(vl + (r0 + xc0) * v_rs);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(vl + (r0 + xc1) * v_rs);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(vl + (r1 + xc0) * v_rs);
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(vl + (r1 + xc1) * v_rs);
            const float ww = inside ? w : 0.f;
            float w1 = ((y0 && x0) ? hh * hw : 0.f) * ww, w2 = ((y0 && x1) ? hh * lw : 0.f) * ww;
            float w3 = ((y1 && x0) ? lh * hw : 0.f) * ww, w4 = ((y1 && x1) ? lh * lw : 0.f) * ww;
            // opaque to the optimiser: otherwise it re-creates a branch around every load whose weight may be zero
            asm volatile("" : "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4));
            msda_fma4(acc, w1, v1);
            msda_fma4(acc, w2, v2);
            msda_fma4(acc, w3, v3);
            msda_fma4(acc, w4, v4);
        }
    }
    *reinterpret_cast<f32x4*>(out + (size_t)q_global * (HEADS * CH) + m * CH + c4) = acc;
}

// The same fused op with the per-sample arithmetic DISTRIBUTED over the eight lanes of a head (round 2).  The kernel above lets
// all eight lanes of a head recompute every sample's location, bilinear weights and corner offsets: rocprofv3 counted 1764
// VALU instructions per query-wave -- 854 us of pure VALU issue per encoder call on each CU, i.e. the whole kernel (930 us);
// the gather itself (64 corner lines per query through the 64 B/clk vector-L1 path) needs ~500 us.  Here lane k of a head owns
// samples 2k and 2k + 1 (level k >> 1): it loads only their offsets and logits, computes their location, validity, the four
// corner weights, the attention weight and the four corner ELEMENT OFFSETS once, and the eight lanes then read each sample's
// nine values from its owner with ds_swizzle (the LDS crossbar: no VALU issue, no memory).  Per-channel arithmetic is the
// kernel above's, value for value: softmax numerators by the same expf, their sum in the same ascending order (every lane
// collects the sixteen numerators by swizzle and adds them itself); from round 6 on a sample is accumulated as four fused
// multiply-adds with corner weights that already carry the sample's attention weight (msda_fma4 above), in every fused kernel.
// Corner loads are buffer loads with 32-bit byte offsets from the batch image (a level map holds < 2^29 floats).
__device__ __forceinline__ float group8_read(float v, int owner) {      // value of lane (lane & ~7) | owner, owner a constant
    // ds_swizzle bit-mask mode: lane' = ((lane & and_mask) | or_mask) ^ xor_mask inside each half-wave of 32
    switch (owner) {
        case 0: return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x18 | (0 << 5)));
        case 1: return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x18 | (1 << 5)));
        case 2: return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x18 | (2 << 5)));
        case 3: return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x18 | (3 << 5)));
        case 4: return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x18 | (4 << 5)));
        case 5: return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x18 | (5 << 5)));
        case 6: return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x18 | (6 << 5)));
        default: return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x18 | (7 << 5)));
    }
}


template <bool HAS_VR>
__global__ __launch_bounds__(256) void msda_fused_lanes_kernel(const float* __restrict__ value,
                                                               const int64_t* __restrict__ shapes,
                                                               const int64_t* __restrict__ lsi,
                                                               const float* __restrict__ raw, int ld_raw,
                                                               const float* __restrict__ ref, float* __restrict__ out,
                                                               int B, int Lq, long v_bs, int v_rs,
                                                               const float* __restrict__ vr, int q_begin, int q_count) {
    constexpr int POINTS = 4, LP = LEVELS * POINTS;
    // queries [q_begin, q_begin + q_count) of every frame (the whole frame by default; the tail behind the windowed kernel's part)
    const long q_local = xcd_contiguous_block(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    if (q_local >= (long)B * q_count) return;
    const int lane = threadIdx.x & 63;
    const int m = lane >> 3, k = lane & 7;
    const int b = (int)(q_local / q_count);
    const long q_global = (long)b * Lq + q_begin + (q_local - (long)b * q_count);
    const int l = k >> 1;                                    // level of this lane's two samples (2k, 2k + 1)

    // ---- owner part: two samples per lane ----
    const float* op = raw + (size_t)q_global * ld_raw + m * (LP * 2) + 4 * k;      // (ox, oy) of samples 2k, 2k + 1
    const float* lp = raw + (size_t)q_global * ld_raw + HEADS * LP * 2 + m * LP + 2 * k;
    const f32x4 off = *reinterpret_cast<const f32x4*>(op);
    const float lg0 = lp[0], lg1 = lp[1];
    const float rx = ref[q_global * 2], ry = ref[q_global * 2 + 1];
    // softmax over the head's 16 logits: max over the eight lanes' pairs (order-free), numerators by the same expf, the sum in
    // ascending sample order on every lane (the fused kernel above sums e[0] .. e[15] in that order)
    float mx = fmaxf(lg0, lg1);
    mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, mx), 0x1F | (1 << 10))));
    mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, mx), 0x1F | (2 << 10))));
    mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, mx), 0x1F | (4 << 10))));
    const float e0 = expf(lg0 - mx), e1 = expf(lg1 - mx);
    float sum = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        sum += group8_read(e0, o);
        sum += group8_read(e1, o);
    }
    const float inv_sum = 1.f / sum;

    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const float Hf = (float)H, Wf = (float)W;
    const float rW = 1.f / Wf, rH = 1.f / Hf;
    const unsigned lvl = (unsigned)lsi[l];
    float sw1[2], sw2[2], sw3[2], sw4[2];                    // corner weights x the sample's attention weight
    unsigned so1[2], so2[2], so3[2], so4[2];                 // BYTE offsets of the four corners' head slice in the batch image
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float ox = off[2 * t], oy = off[2 * t + 1];
        float qx = ox * rW, qy = oy * rH;
        qx = fmaf(fmaf(-qx, Wf, ox), rW, qx);
        qy = fmaf(fmaf(-qy, Hf, oy), rH, qy);
        const float lx = (HAS_VR ? rx * vr[2 * l] : rx) + qx, ly = (HAS_VR ? ry * vr[2 * l + 1] : ry) + qy;
        const float w = (t ? e1 : e0) * inv_sum;
        const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
        const bool inside = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
        const float hf = floorf(h_im), wf = floorf(w_im);
        const float lh = h_im - hf, lw = w_im - wf;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const int h_low = inside ? (int)hf : 0, w_low = inside ? (int)wf : 0;
        const bool y0 = h_low >= 0, y1 = h_low + 1 <= H - 1;
        const bool x0 = w_low >= 0, x1 = w_low + 1 <= W - 1;
        const int yc0 = y0 ? h_low : 0, yc1 = y1 ? h_low + 1 : H - 1;
        const int xc0 = x0 ? w_low : 0, xc1 = x1 ? w_low + 1 : W - 1;
        const unsigned r0 = lvl + (unsigned)(yc0 * W), r1 = lvl + (unsigned)(yc1 * W);
        const unsigned head = (unsigned)(m * CH) * 4u;
        so1[t] = (r0 + xc0) * (unsigned)v_rs * 4u + head;
        so2[t] = (r0 + xc1) * (unsigned)v_rs * 4u + head;
        so3[t] = (r1 + xc0) * (unsigned)v_rs * 4u + head;
        so4[t] = (r1 + xc1) * (unsigned)v_rs * 4u + head;
        const float ww = inside ? w : 0.f;
        sw1[t] = ((y0 && x0) ? hh * hw : 0.f) * ww;
        sw2[t] = ((y0 && x1) ? hh * lw : 0.f) * ww;
        sw3[t] = ((y1 && x0) ? lh * hw : 0.f) * ww;
        sw4[t] = ((y1 && x1) ? lh * lw : 0.f) * ww;
    }

    // ---- gather part: every lane, all 16 samples, 4 channels ----
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(value + (size_t)b * v_bs), 0, 0x7FFFFFFF, 0x00020000);
    const unsigned mine = (unsigned)k * 16u;                 // this lane's 4 channels inside the head's 128-byte slice
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < LP; ++i) {
        const int o = i >> 1, t = i & 1;
        const unsigned a1 = __builtin_bit_cast(unsigned, group8_read(__builtin_bit_cast(float, so1[t]), o)) + mine;
        const unsigned a2 = __builtin_bit_cast(unsigned, group8_read(__builtin_bit_cast(float, so2[t]), o)) + mine;
        const unsigned a3 = __builtin_bit_cast(unsigned, group8_read(__builtin_bit_cast(float, so3[t]), o)) + mine;
        const unsigned a4 = __builtin_bit_cast(unsigned, group8_read(__builtin_bit_cast(float, so4[t]), o)) + mine;
        const f32x4 v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a1, 0, 0));
        const f32x4 v2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a2, 0, 0));
        const f32x4 v3 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a3, 0, 0));
        const f32x4 v4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a4, 0, 0));
        const float w1 = group8_read(sw1[t], o), w2 = group8_read(sw2[t], o), w3 = group8_read(sw3[t], o), w4 = group8_read(sw4[t], o);
        msda_fma4(acc, w1, v1);
        msda_fma4(acc, w2, v2);
        msda_fma4(acc, w3, v3);
        msda_fma4(acc, w4, v4);
    }
    *reinterpret_cast<f32x4*>(out + (size_t)q_global * (HEADS * CH) + m * CH + k * 4) = acc;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The fused op for the ENCODER's level-0 queries with the value map served from LDS (round 4).
//
// The kernels above are bound by the texture-address path: 512 scattered 128-byte corner lines per query at ~3.6 cycles each
// (profiles/r03_msda_ta_counters.txt; no gather shape or value layout is cheaper: profiles/r04_msda_ta_counters.txt).  But an
// encoder query IS a pixel, and its 128 samples fall within a few pixels of that pixel's position on every level: the raw
// offsets are in pixels of the sampled level (loc = ref + off / (W_l, H_l)), |off| <= 5.4 over six layers of the bench
// workload, p99.9 = 4 (tools/diag/msda_offsets.py; the reference initialises the offsets' bias to k = 1..4 pixels along eight
// directions).  So a workgroup that owns a TILE of TY x TX level-0 queries and ONE head needs, per level, only the tile's
// projection plus a halo of R pixels: it loads those lines ONCE (whole 128-byte lines, ~12 per (query, head) instead of 64
// gathered ones) into LDS and serves every corner from there with ds_read_b128.
//   * one workgroup = (frame, tile, head), 4 waves; a wave handles 8 (query, head) pairs at a time (8 lanes each, 4 channels per
//     lane -- the lane-distributed kernel's layout with the eight heads of a wave replaced by eight queries), 4 such octet
//     groups per wave = 128 queries per tile;
//   * the per-sample arithmetic is the lane-distributed kernel's, instruction for instruction (softmax, location, the four
//     corner weights x the attention weight, four fused multiply-adds per sample in sample order), so results are BIT-IDENTICAL;
//   * levels are processed one after the other through one LDS buffer (fill level l, barrier, its four samples of every query);
//     the accumulation order l = 0..3, p = 0..3 is the kernels' above;
//   * a sample whose clamped corners are not all inside the window makes its wave's octet group (8 queries) take the global-
//     memory path for that group -- the lane-distributed kernel's arithmetic again, same bits; never taken on the bench workload;
//   * workgroup id = tile * 8 + head: workgroups go round-robin to the 8 XCDs, so XCD m sees exactly head m's lines (1/8 of the
//     value map per L2) and consecutive tiles, which share their halos, meet in the same L2.
// Queries of the coarser levels (25 % of the tokens) keep the lane-distributed kernel (q_begin / q_count above).
// GLV (round 6): the samples of the GLV finest levels are GATHERED from global memory (the lane-distributed kernel's loads, its
// arithmetic, its accumulation order) instead of served from a window -- for tiles of level-1 queries, whose projection on level 0
// (twice the pixels per query) is the window that does not fit: an 8 x 16 level-1 tile needs 27 x 43 = 1 161 level-0 lines, its
// level-1 .. 3 windows 513 + 285 + 195.
template <int TY, int TX, int R, int CAP, int NB, int GLV = 0>
__global__ __launch_bounds__(256, 2) void msda_window_kernel(const float* __restrict__ value,
                                                             const int64_t* __restrict__ shapes,
                                                             const int64_t* __restrict__ lsi,
                                                             const float* __restrict__ raw, int ld_raw,
                                                             const float* __restrict__ ref, float* __restrict__ out,
                                                             int Lq, long v_bs, int v_rs, int tiles_y, int tiles_x, int ql,
                                                             unsigned* __restrict__ slow_groups) {
    constexpr int POINTS = 4, LP = LEVELS * POINTS, ITERS = (TY * TX) / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char win[];          // CAP lines of 128 bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 3, k = lane & 7;
    const int m = (int)(blockIdx.x & 7u);                    // head
    unsigned tile = blockIdx.x >> 3;
    const int tx = (int)(tile % (unsigned)tiles_x);
    tile /= (unsigned)tiles_x;
    const int ty = (int)(tile % (unsigned)tiles_y);
    const int b = (int)(tile / (unsigned)tiles_y);
    const int l = k >> 1;                                    // level of this lane's two samples (2k, 2k + 1)

    int Hs[LEVELS], Ws[LEVELS];
    unsigned lv[LEVELS];
#pragma unroll
    for (int i = 0; i < LEVELS; ++i) {
        Hs[i] = (int)shapes[2 * i];
        Ws[i] = (int)shapes[2 * i + 1];
        lv[i] = (unsigned)lsi[i];
    }
    // the tile's queries are pixels of level `ql` (0: the finest map; 1: round 6, smaller tiles -- the same windows cover twice the
    // pixels per query on the finer levels)
    const int Hq = ql == 0 ? Hs[0] : ql == 1 ? Hs[1] : ql == 2 ? Hs[2] : Hs[3];
    const int Wq = ql == 0 ? Ws[0] : ql == 1 ? Ws[1] : ql == 2 ? Ws[2] : Ws[3];
    const long q0 = (long)b * Lq + (long)(ql == 0 ? lv[0] : ql == 1 ? lv[1] : ql == 2 ? lv[2] : lv[3]);
    const int y0t = ty * TY, x0t = tx * TX;
    const int y1t = min(y0t + TY, Hq) - 1, x1t = min(x0t + TX, Wq) - 1;
    // windows: the projection of the tile's first and last query on every level, R pixels of halo + the second bilinear corner
    const long qa = q0 + (long)y0t * Wq + x0t, qz = q0 + (long)y1t * Wq + x1t;
    const float rxa = ref[qa * 2], rya = ref[qa * 2 + 1], rxz = ref[qz * 2], ryz = ref[qz * 2 + 1];
    int wx0[LEVELS], wy0[LEVELS], wx1[LEVELS], wy1[LEVELS], wwd[LEVELS];
#pragma unroll
    for (int i = 0; i < LEVELS; ++i) {
        int ax = (int)floorf(fminf(rxa, rxz) * Ws[i] - 0.5f) - R, zx = (int)floorf(fmaxf(rxa, rxz) * Ws[i] - 0.5f) + R + 1;
        int ay = (int)floorf(fminf(rya, ryz) * Hs[i] - 0.5f) - R, zy = (int)floorf(fmaxf(rya, ryz) * Hs[i] - 0.5f) + R + 1;
        ax = max(ax, 0); ay = max(ay, 0); zx = min(zx, Ws[i] - 1); zy = min(zy, Hs[i] - 1);
        if (zx < ax) zx = ax;
        if (zy < ay) zy = ay;
        int w_ = zx - ax + 1, h_ = zy - ay + 1;
        if (w_ > CAP) { w_ = CAP; zx = ax + w_ - 1; }
        if (w_ * h_ > CAP) { h_ = CAP / w_; zy = ay + h_ - 1; }   // whatever does not fit is served by the global path
        wx0[i] = ax; wy0[i] = ay; wx1[i] = zx; wy1[i] = zy; wwd[i] = w_;
    }
    // this lane's level (runtime index l): select with compares, not an indexed array (registers)
    const int H = l == 0 ? Hs[0] : l == 1 ? Hs[1] : l == 2 ? Hs[2] : Hs[3];
    const int W = l == 0 ? Ws[0] : l == 1 ? Ws[1] : l == 2 ? Ws[2] : Ws[3];
    const int mx0 = l == 0 ? wx0[0] : l == 1 ? wx0[1] : l == 2 ? wx0[2] : wx0[3];
    const int my0 = l == 0 ? wy0[0] : l == 1 ? wy0[1] : l == 2 ? wy0[2] : wy0[3];
    const int mx1 = l == 0 ? wx1[0] : l == 1 ? wx1[1] : l == 2 ? wx1[2] : wx1[3];
    const int my1 = l == 0 ? wy1[0] : l == 1 ? wy1[1] : l == 2 ? wy1[2] : wy1[3];
    const int mww = l == 0 ? wwd[0] : l == 1 ? wwd[1] : l == 2 ? wwd[2] : wwd[3];
    const unsigned lvl = l == 0 ? lv[0] : l == 1 ? lv[1] : l == 2 ? lv[2] : lv[3];
    const float Hf = (float)H, Wf = (float)W;
    const float rW = 1.f / Wf, rH = 1.f / Hf;

    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(value + (size_t)b * v_bs), 0, 0x7FFFFFFF, 0x00020000);
    const unsigned mine = (unsigned)k * 16u;                 // this lane's 4 channels inside a 128-byte line
    const unsigned head = (unsigned)(m * CH) * 4u;

    // ---- owner part of every octet group: the lane-distributed kernel's, plus the window test and the LDS address ----
    // (straight-line code over the ITERS groups: no branch in here or in the level loops, so that the scheduler can overlap the
    // groups' loads, swizzles and LDS reads -- a wave has at most one partner on its SIMD to hide latency behind)
    float sw1[ITERS][2], sw2[ITERS][2], sw3[ITERS][2], sw4[ITERS][2];   // corner weights x the sample's attention weight
    unsigned pk[ITERS][2];                                   // LDS byte address of corner (yc0, xc0) | dx << 20 | dy << 21
    unsigned go1[GLV ? ITERS : 1][2], go2[GLV ? ITERS : 1][2], go3[GLV ? ITERS : 1][2], go4[GLV ? ITERS : 1][2];   // GLV: global corner offsets
    f32x4 acc[ITERS];
    long qrow[ITERS];
    bool fast[ITERS], live[ITERS];
    f32x4 offs[ITERS];
    float lga[ITERS], lgb[ITERS], rxs[ITERS], rys[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int j = (wave * ITERS + it) * 8 + g;           // query of the tile
        const int jy = j / TX, jx = j - jy * TX;
        const int qy = y0t + jy, qx = x0t + jx;
        live[it] = qy <= y1t && qx <= x1t;
        const long q_global = q0 + (long)(live[it] ? qy : y0t) * Wq + (live[it] ? qx : x0t);
        qrow[it] = q_global;
        const float* op = raw + (size_t)q_global * ld_raw + m * (LP * 2) + 4 * k;
        const float* lp = raw + (size_t)q_global * ld_raw + HEADS * LP * 2 + m * LP + 2 * k;
        offs[it] = *reinterpret_cast<const f32x4*>(op);
        lga[it] = lp[0];
        lgb[it] = lp[1];
        rxs[it] = ref[q_global * 2];
        rys[it] = ref[q_global * 2 + 1];
    }
    // geometry of sample (IT, T) of this lane (a macro, not a lambda: the arrays must be indexed by literals to stay in registers)
#define MSDA_GEOMETRY(IT, T, OFF4, RX, RY, E, INV_SUM, OK, SO)                                                    \
    {                                                                                                             \
        const float ox = (OFF4)[2 * (T)], oy = (OFF4)[2 * (T) + 1];                                               \
        float qx_ = ox * rW, qy_ = oy * rH;                                                                       \
        qx_ = fmaf(fmaf(-qx_, Wf, ox), rW, qx_);                                                                  \
        qy_ = fmaf(fmaf(-qy_, Hf, oy), rH, qy_);                                                                  \
        const float lx = (RX) + qx_, ly = (RY) + qy_;                                                             \
        const float w = (E) * (INV_SUM);                                                                          \
        const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;                                                   \
        const bool inside = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;                                 \
        const float hf = floorf(h_im), wf = floorf(w_im);                                                         \
        const float lh = h_im - hf, lw = w_im - wf;                                                               \
        const float hh = 1.f - lh, hw = 1.f - lw;                                                                 \
        const int h_low = inside ? (int)hf : 0, w_low = inside ? (int)wf : 0;                                     \
        const bool y0 = h_low >= 0, y1 = h_low + 1 <= H - 1;                                                      \
        const bool x0 = w_low >= 0, x1 = w_low + 1 <= W - 1;                                                      \
        const int yc0 = y0 ? h_low : 0, yc1 = y1 ? h_low + 1 : H - 1;                                             \
        const int xc0 = x0 ? w_low : 0, xc1 = x1 ? w_low + 1 : W - 1;                                             \
        const unsigned r0 = lvl + (unsigned)(yc0 * W), r1 = lvl + (unsigned)(yc1 * W);                            \
        SO##1 = (r0 + xc0) * (unsigned)v_rs * 4u + head;                                                          \
        SO##2 = (r0 + xc1) * (unsigned)v_rs * 4u + head;                                                          \
        SO##3 = (r1 + xc0) * (unsigned)v_rs * 4u + head;                                                          \
        SO##4 = (r1 + xc1) * (unsigned)v_rs * 4u + head;                                                          \
        const float ww_ = inside ? w : 0.f;                                                                       \
        sw1[IT][T] = ((y0 && x0) ? hh * hw : 0.f) * ww_;                                                          \
        sw2[IT][T] = ((y0 && x1) ? hh * lw : 0.f) * ww_;                                                          \
        sw3[IT][T] = ((y1 && x0) ? lh * hw : 0.f) * ww_;                                                          \
        sw4[IT][T] = ((y1 && x1) ? lh * lw : 0.f) * ww_;                                                          \
        /* a sample outside the map carries weight 0: any finite line serves (the kernels above read the map's corner there) */ \
        OK = !inside || l < GLV || (yc0 >= my0 && yc1 <= my1 && xc0 >= mx0 && xc1 <= mx1);                        \
        const bool use = inside && OK; /* (an address inside the buffer in every case) */                         \
        const int ly0 = use ? yc0 - my0 : 0, lx0 = use ? xc0 - mx0 : 0;                                           \
        const unsigned la_ = (unsigned)((ly0 * mww + lx0) * 128);                                                 \
        pk[IT][T] = la_ | ((use && xc1 != xc0) ? (1u << 20) : 0u) | ((use && yc1 != yc0) ? (1u << 21) : 0u);      \
    }
    unsigned any_slow = 0;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const float lg0 = lga[it], lg1 = lgb[it];
        float mx = fmaxf(lg0, lg1);
        mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, mx), 0x1F | (1 << 10))));
        mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, mx), 0x1F | (2 << 10))));
        mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, mx), 0x1F | (4 << 10))));
        const float e0 = expf(lg0 - mx), e1 = expf(lg1 - mx);
        float sum = 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            sum += group8_read(e0, o);
            sum += group8_read(e1, o);
        }
        const float inv_sum = 1.f / sum;
        bool ok0, ok1;
        unsigned sa1, sa2, sa3, sa4, sb1, sb2, sb3, sb4;
        MSDA_GEOMETRY(it, 0, offs[it], rxs[it], rys[it], e0, inv_sum, ok0, sa)
        MSDA_GEOMETRY(it, 1, offs[it], rxs[it], rys[it], e1, inv_sum, ok1, sb)
        if constexpr (GLV > 0) {                                // the global byte offsets of this lane's two samples' corners
            go1[it][0] = sa1; go2[it][0] = sa2; go3[it][0] = sa3; go4[it][0] = sa4;
            go1[it][1] = sb1; go2[it][1] = sb2; go3[it][1] = sb3; go4[it][1] = sb4;
        }
        (void)sa1; (void)sa2; (void)sa3; (void)sa4; (void)sb1; (void)sb2; (void)sb3; (void)sb4;
        fast[it] = __builtin_amdgcn_ballot_w64(!(ok0 && ok1)) == 0;   // wave-uniform: every sample of the 8 queries is inside
        any_slow |= fast[it] ? 0u : 1u;
        acc[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- level by level: fill the window, then the level's four samples of every octet group ----
    // Fill = LDS-DMA (buffer_load_dwordx4 ... lds): a wave-instruction moves 8 lines (1 KB) into 8 consecutive window lines,
    // every lane from its own source address; all requests of a level are in flight together.  (Rounds 4-5 also carried a double-
    // buffered form -- the next level's fill under this level's samples, one workgroup per CU -- and forms with two / four corner
    // addresses computed at the owner, a DPP broadcast: none faster, tools/exp + docs/LAB_NOTES.md; removed in round 6.)
    // (the level is a compile-time constant of every call: as a run-time loop the compiler keeps `lev` in a register, indexes the
    // windows' arrays in scratch and turns every literal owner of a broadcast into an eight-way branch)
    auto fill = [&](auto LEVC, unsigned char* buf) {
        constexpr int lev = decltype(LEVC)::value;
        const int ww_ = wwd[lev], n_lines = ww_ * (wy1[lev] - wy0[lev] + 1);
        const float inv_w = 1.f / (float)ww_;
        for (int base = wave * 8; base < n_lines; base += 32) {
            int line = base + g;
            if (line > n_lines - 1) line = n_lines - 1;      // the tail of the last group re-reads the last line
            int yy = (int)(((float)line + 0.5f) * inv_w);
            int xx = line - yy * ww_;
            if (xx < 0) { --yy; xx += ww_; }
            if (xx >= ww_) { ++yy; xx -= ww_; }
            const unsigned src = (lv[lev] + (unsigned)((wy0[lev] + yy) * Ws[lev] + wx0[lev] + xx)) * (unsigned)v_rs * 4u + head + mine;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(buf + (size_t)base * 128), 16,
                                                     (int)src, 0, 0, 0);
        }
    };
    auto bcast = [](float v, int owner) { return group8_read(v, owner); };
    auto level_samples = [&](auto LEVC, const unsigned char* buf) {
        constexpr int lev = decltype(LEVC)::value;
        typedef __attribute__((address_space(3))) f32x4 LdsF4;
        const unsigned mine_lds = mine + (unsigned)(size_t)buf;   // (a flat LDS address: its low word is the LDS offset)
        const unsigned row_bytes = (unsigned)wwd[lev] * 128u;
#pragma unroll
        for (int pnt = 0; pnt < POINTS; ++pnt) {
            const int i = lev * POINTS + pnt, o = i >> 1, t = i & 1;
#pragma unroll
            for (int half = 0; half < ITERS; half += NB) {    // NB octet groups at a time: their swizzles, then their reads, then the FMAs
                // (wave-uniform, almost never taken -- and it keeps every batch a basic block of its own: as straight-line code
                // the compiler computes the swizzles of ALL levels in front of the first barrier and spills 3.6 KB per lane)
                bool any_fast = false;
#pragma unroll
                for (int u = 0; u < NB; ++u) any_fast = any_fast || fast[half + u];
                if (!any_fast) continue;
                unsigned word[NB];
                float w1[NB], w2[NB], w3[NB], w4[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int it = half + u;
                    // (opaque: a ds_swizzle is not a memory operation, and without this the compiler computes the swizzles of ALL
                    // levels in front of the first barrier -- 1500 live values, 3.6 KB of scratch per lane)
                    asm volatile("" : "+v"(pk[it][t]), "+v"(sw1[it][t]), "+v"(sw2[it][t]), "+v"(sw3[it][t]), "+v"(sw4[it][t]));
                    word[u] = __builtin_bit_cast(unsigned, bcast(__builtin_bit_cast(float, pk[it][t]), o));
                    w1[u] = bcast(sw1[it][t], o);
                    w2[u] = bcast(sw2[it][t], o);
                    w3[u] = bcast(sw3[it][t], o);
                    w4[u] = bcast(sw4[it][t], o);
                }
                f32x4 v1[NB], v2[NB], v3[NB], v4[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const unsigned a1 = (word[u] & 0xFFFFFu) + mine_lds;
                    const unsigned dx = (word[u] >> 20) & 1u, dy = (word[u] >> 21) & 1u;
                    unsigned a2 = a1 + dx * 128u, a3 = a1 + dy * row_bytes, a4 = a3 + dx * 128u;
                    // opaque: otherwise the compiler branches around the reads whose address may equal another one's (dx = 0 or
                    // dy = 0 at the map's edge) -- 600 branches in this kernel, exec-masked paths per lane group
                    asm volatile("" : "+v"(a2), "+v"(a3), "+v"(a4));
                    // (absolute LDS addresses -- the window's offset rides in `mine_lds`: as `buf + a` behind the opaque asm the
                    // compiler spends one v_add_u32 per read on a base it cannot fold, 4 of ~30 vector instructions per sample)
                    v1[u] = *reinterpret_cast<const LdsF4*>((size_t)a1);
                    v2[u] = *reinterpret_cast<const LdsF4*>((size_t)a2);
                    v3[u] = *reinterpret_cast<const LdsF4*>((size_t)a3);
                    v4[u] = *reinterpret_cast<const LdsF4*>((size_t)a4);
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    msda_fma4(acc[half + u], w1[u], v1[u]);
                    msda_fma4(acc[half + u], w2[u], v2[u]);
                    msda_fma4(acc[half + u], w3[u], v3[u]);
                    msda_fma4(acc[half + u], w4[u], v4[u]);
                }
                __builtin_amdgcn_sched_barrier(0);           // (keeps the scheduler from hoisting later batches' loads: spills)
            }
        }
    };
#define MSDA_LEVEL_SINGLE(L)                                                                                      \
    {                                                                                                             \
        __syncthreads(); /* the previous level's reads are done */                                                \
        fill(std::integral_constant<int, L>{}, win);                                                              \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
        __syncthreads();                                                                                          \
        level_samples(std::integral_constant<int, L>{}, win);                                                     \
    }
    // GLV: the four samples of a GATHERED level -- the lane-distributed kernel's inner loop (corner offsets broadcast from the owner
    // lane, four 16-byte loads of the head's 128-byte slices, the same products in the same order)
    auto level_gather = [&](auto LEVC) {
        constexpr int lev = decltype(LEVC)::value;
#pragma unroll
        for (int pnt = 0; pnt < POINTS; ++pnt) {
            const int i = lev * POINTS + pnt, o = i >> 1, t = i & 1;
#pragma unroll
            for (int half = 0; half < ITERS; half += NB) {
                bool any_fast = false;
#pragma unroll
                for (int u = 0; u < NB; ++u) any_fast = any_fast || fast[half + u];
                if (!any_fast) continue;
                unsigned a1[NB], a2[NB], a3[NB], a4[NB];
                float w1[NB], w2[NB], w3[NB], w4[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int it = half + u;
                    asm volatile("" : "+v"(go1[it][t]), "+v"(go2[it][t]), "+v"(go3[it][t]), "+v"(go4[it][t]));
                    asm volatile("" : "+v"(sw1[it][t]), "+v"(sw2[it][t]), "+v"(sw3[it][t]), "+v"(sw4[it][t]));
                    a1[u] = __builtin_bit_cast(unsigned, bcast(__builtin_bit_cast(float, go1[it][t]), o)) + mine;
                    a2[u] = __builtin_bit_cast(unsigned, bcast(__builtin_bit_cast(float, go2[it][t]), o)) + mine;
                    a3[u] = __builtin_bit_cast(unsigned, bcast(__builtin_bit_cast(float, go3[it][t]), o)) + mine;
                    a4[u] = __builtin_bit_cast(unsigned, bcast(__builtin_bit_cast(float, go4[it][t]), o)) + mine;
                    w1[u] = bcast(sw1[it][t], o);
                    w2[u] = bcast(sw2[it][t], o);
                    w3[u] = bcast(sw3[it][t], o);
                    w4[u] = bcast(sw4[it][t], o);
                }
                f32x4 v1[NB], v2[NB], v3[NB], v4[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    v1[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a1[u], 0, 0));
                    v2[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a2[u], 0, 0));
                    v3[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a3[u], 0, 0));
                    v4[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a4[u], 0, 0));
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    msda_fma4(acc[half + u], w1[u], v1[u]);
                    msda_fma4(acc[half + u], w2[u], v2[u]);
                    msda_fma4(acc[half + u], w3[u], v3[u]);
                    msda_fma4(acc[half + u], w4[u], v4[u]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    static_assert(GLV == 0 || GLV == 1, "only the finest level is ever gathered");
    if constexpr (GLV == 0) {
        MSDA_LEVEL_SINGLE(0)
    } else {
        // level 1's window is requested first and lands under level 0's gathers (accumulation order l = 0, 1, 2, 3 as everywhere)
        fill(std::integral_constant<int, 1>{}, win);
        level_gather(std::integral_constant<int, 0>{});
    }
    if constexpr (GLV == 0) {
        MSDA_LEVEL_SINGLE(1)
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        level_samples(std::integral_constant<int, 1>{}, win);
    }
    MSDA_LEVEL_SINGLE(2) MSDA_LEVEL_SINGLE(3)
#undef MSDA_LEVEL_SINGLE
    if (any_slow) {
        if (slow_groups && lane == 0) {                       // diagnostic count (gom_msda_window_count_fallbacks): octet groups on the global path
            unsigned n_slow = 0;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) n_slow += fast[it] ? 0u : 1u;
            atomicAdd(slow_groups, n_slow);
        }
        // octet groups with a sample outside its window: the lane-distributed kernel's gather from global memory (rare: never on
        // the bench workload); the window result of such a group is discarded
#pragma unroll                                                // (static indices: a run-time `it` would put the arrays in scratch)
        for (int it = 0; it < ITERS; ++it) {
            if (fast[it]) continue;
            const long q_global = qrow[it];
            const f32x4 off4 = *reinterpret_cast<const f32x4*>(raw + (size_t)q_global * ld_raw + m * (LP * 2) + 4 * k);
            const float* lp = raw + (size_t)q_global * ld_raw + HEADS * LP * 2 + m * LP + 2 * k;
            const float lg0 = lp[0], lg1 = lp[1];
            float mx = fmaxf(lg0, lg1);
            mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, mx), 0x1F | (1 << 10))));
            mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, mx), 0x1F | (2 << 10))));
            mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, mx), 0x1F | (4 << 10))));
            const float e0 = expf(lg0 - mx), e1 = expf(lg1 - mx);
            float sum = 0.f;
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                sum += group8_read(e0, o);
                sum += group8_read(e1, o);
            }
            const float inv_sum = 1.f / sum;
            bool ok_;
            unsigned sa1, sa2, sa3, sa4, sb1, sb2, sb3, sb4;
            const float rx_ = ref[q_global * 2], ry_ = ref[q_global * 2 + 1];
            MSDA_GEOMETRY(it, 0, off4, rx_, ry_, e0, inv_sum, ok_, sa)
            MSDA_GEOMETRY(it, 1, off4, rx_, ry_, e1, inv_sum, ok_, sb)
            (void)ok_;
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < LP; ++i) {
                const int o = i >> 1, t = i & 1;
                const unsigned a1 = __builtin_bit_cast(unsigned, group8_read(__builtin_bit_cast(float, t ? sb1 : sa1), o)) + mine;
                const unsigned a2 = __builtin_bit_cast(unsigned, group8_read(__builtin_bit_cast(float, t ? sb2 : sa2), o)) + mine;
                const unsigned a3 = __builtin_bit_cast(unsigned, group8_read(__builtin_bit_cast(float, t ? sb3 : sa3), o)) + mine;
                const unsigned a4 = __builtin_bit_cast(unsigned, group8_read(__builtin_bit_cast(float, t ? sb4 : sa4), o)) + mine;
                const f32x4 v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a1, 0, 0));
                const f32x4 v2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a2, 0, 0));
                const f32x4 v3 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a3, 0, 0));
                const f32x4 v4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a4, 0, 0));
                const float w1 = group8_read(sw1[it][t], o), w2 = group8_read(sw2[it][t], o), w3 = group8_read(sw3[it][t], o),
                            w4 = group8_read(sw4[it][t], o);
                msda_fma4(r, w1, v1);
                msda_fma4(r, w2, v2);
                msda_fma4(r, w3, v3);
                msda_fma4(r, w4, v4);
            }
            acc[it] = r;
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
        if (live[it]) *reinterpret_cast<f32x4*>(out + (size_t)qrow[it] * (HEADS * CH) + m * CH + k * 4) = acc[it];
}

#undef MSDA_GEOMETRY

}  // namespace

static int g_msda_lanes = 1;
/* [host] 1 (default): the lane-distributed fused kernel; 0: the one-lane-does-all form (A/B runs, tests). */
extern "C" int gom_msda_set_lane_distributed(int on) {
    g_msda_lanes = on ? 1 : 0;
    return GOM_OK;
}

extern "C" int gom_msda_fused_forward(const float* raw, int ld_raw, const float* ref, const float* value,
                                      long value_batch_stride, int value_row_stride, const int64_t* spatial_shapes,
                                      const int64_t* level_start_index, float* output, int batch, int num_query,
                                      void* stream) {
    GOM_CHECK_ARG(raw && ref && value && spatial_shapes && level_start_index && output);
    GOM_CHECK_ARG(batch > 0 && num_query > 0 && ld_raw >= HEADS * LEVELS * 4 * 3 && (ld_raw % 4) == 0);
    GOM_CHECK_ARG(value_row_stride >= HEADS * CH && (value_row_stride % 4) == 0 && (value_batch_stride % 4) == 0);
    GOM_CHECK_ARG(((uintptr_t)raw % 16) == 0 && ((uintptr_t)value % 16) == 0);
    const long nq = (long)batch * num_query;
    if (g_msda_lanes && value_batch_stride > 0 && value_batch_stride < (1L << 29))     // 32-bit byte offsets inside a batch image
        hipLaunchKernelGGL((msda_fused_lanes_kernel<false>), dim3((unsigned)cdiv(nq, 4)), dim3(256), 0, (hipStream_t)stream,
                           value, spatial_shapes, level_start_index, raw, ld_raw, ref, output, batch, num_query,
                           value_batch_stride, value_row_stride, (const float*)nullptr, 0, num_query);
    else
        hipLaunchKernelGGL((msda_fused_kernel<4, false>), dim3((unsigned)cdiv(nq, 4)), dim3(256), 0, (hipStream_t)stream, value,
                           spatial_shapes, level_start_index, raw, ld_raw, ref, output, batch, num_query,
                           value_batch_stride, value_row_stride, (const float*)nullptr);
    return gom_launch_status();
}

static int g_msda_window = 3;
static unsigned* g_msda_slow = nullptr;
/* [host] which encoder queries the entry below serves from LDS windows: bit 0 = the level-0 pixels (8 x 16 tiles, round 4), bit 1 =
 * the level-1 pixels (4 x 8 tiles, round 6); default 3 = both, 0 = everything on the lane-distributed kernel (A/B runs, tests).
 * Same bits either way. */
extern "C" int gom_msda_set_window(int mask) {
    g_msda_window = mask < 0 ? 0 : (mask & 3);
    return GOM_OK;
}
/* [host] diagnostic: a device word that every window workgroup adds its FALLBACK octet groups to (groups of 8 (query, head) pairs
 * with a sample outside its window: they take the gather path); NULL (default) = no counting.  tools/msda_offset_scale.py. */
extern "C" int gom_msda_window_count_fallbacks(unsigned int* device_counter) {
    g_msda_slow = device_counter;
    return GOM_OK;
}

/* The fused op for an ENCODER call: num_query = the tokens of the pyramid (query q of a frame IS token q: level-0 pixels first,
 * `h0` x `w0` of them in raster order, then the coarser levels'), reference points = the pixels' own positions.  Level-0 and
 * level-1 queries run on the LDS-window kernel, the coarser levels' on the lane-distributed one; results are bit-identical to
 * gom_msda_fused_forward. */
extern "C" int gom_msda_fused_forward_encoder(const float* raw, int ld_raw, const float* ref, const float* value,
                                              long value_batch_stride, int value_row_stride, const int64_t* spatial_shapes,
                                              const int64_t* level_start_index, float* output, int batch, int num_query,
                                              int h0, int w0, int h1, int w1, void* stream) {
    GOM_CHECK_ARG(raw && ref && value && spatial_shapes && level_start_index && output);
    GOM_CHECK_ARG(batch > 0 && num_query > 0 && ld_raw >= HEADS * LEVELS * 4 * 3 && (ld_raw % 4) == 0);
    GOM_CHECK_ARG(value_row_stride >= HEADS * CH && (value_row_stride % 4) == 0 && (value_batch_stride % 4) == 0);
    GOM_CHECK_ARG(((uintptr_t)raw % 16) == 0 && ((uintptr_t)value % 16) == 0);
    GOM_CHECK_ARG(h0 > 0 && w0 > 0 && h1 >= 0 && w1 >= 0 && (long)h0 * w0 + (long)h1 * w1 <= num_query);
    constexpr int R = 5, CAP = 576;                          // 72 KB of window lines: two workgroups per CU
    constexpr int TY0 = 8, TX0 = 16, TY1 = 8, TX1 = 16;       // level-1 tiles: their level-0 samples are gathered (GLV = 1)
    const long n0 = (long)h0 * w0, n1 = (long)h1 * w1;       // (host copies of spatial_shapes[0], [1]: the tile grids)
    const long wgs0 = (long)batch * cdiv(h0, TY0) * cdiv(w0, TX0) * 8;
    const long wgs1 = (long)batch * cdiv(h1, TY1) * cdiv(w1, TX1) * 8;
    if (!((g_msda_window & 1) && g_msda_lanes && value_batch_stride > 0 && value_batch_stride < (1L << 29) && wgs0 < (1L << 31) &&
          wgs1 < (1L << 31)))
        return gom_msda_fused_forward(raw, ld_raw, ref, value, value_batch_stride, value_row_stride, spatial_shapes,
                                      level_start_index, output, batch, num_query, stream);
    hipStream_t s = (hipStream_t)stream;
    {
        auto kern = msda_window_kernel<TY0, TX0, R, CAP, 4>;
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CAP * 128);
        if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
        hipLaunchKernelGGL(kern, dim3((unsigned)wgs0), dim3(256), CAP * 128, s, value, spatial_shapes, level_start_index, raw, ld_raw,
                           ref, output, num_query, value_batch_stride, value_row_stride, cdiv(h0, TY0), cdiv(w0, TX0), 0, g_msda_slow);
    }
    long done = n0;
    if ((g_msda_window & 2) && n1 > 0) {
        auto kern = msda_window_kernel<TY1, TX1, R, CAP, 4, 1>;
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CAP * 128);
        if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
        hipLaunchKernelGGL(kern, dim3((unsigned)wgs1), dim3(256), CAP * 128, s, value, spatial_shapes, level_start_index, raw, ld_raw,
                           ref, output, num_query, value_batch_stride, value_row_stride, cdiv(h1, TY1), cdiv(w1, TX1), 1, g_msda_slow);
        done += n1;
    }
    const long rest = num_query - done;
    if (rest > 0)                                            // the coarser levels' queries: the lane-distributed kernel (disjoint rows)
        hipLaunchKernelGGL((msda_fused_lanes_kernel<false>), dim3((unsigned)cdiv((long)batch * rest, 4)), dim3(256), 0, s, value,
                           spatial_shapes, level_start_index, raw, ld_raw, ref, output, batch, num_query, value_batch_stride,
                           value_row_stride, (const float*)nullptr, (int)done, (int)rest);
    return gom_launch_status();
}

extern "C" int gom_msda_fused_forward_vr(const float* raw, int ld_raw, const float* ref, const float* value,
                                         long value_batch_stride, int value_row_stride, const int64_t* spatial_shapes,
                                         const int64_t* level_start_index, const float* valid_ratios, float* output,
                                         int batch, int num_query, void* stream) {
    GOM_CHECK_ARG(raw && ref && value && spatial_shapes && level_start_index && valid_ratios && output);
    GOM_CHECK_ARG(batch > 0 && num_query > 0 && ld_raw >= HEADS * LEVELS * 4 * 3 && (ld_raw % 4) == 0);
    GOM_CHECK_ARG(value_row_stride >= HEADS * CH && (value_row_stride % 4) == 0 && (value_batch_stride % 4) == 0);
    const long nq = (long)batch * num_query;
    if (g_msda_lanes && value_batch_stride > 0 && value_batch_stride < (1L << 29))
        hipLaunchKernelGGL((msda_fused_lanes_kernel<true>), dim3((unsigned)cdiv(nq, 4)), dim3(256), 0, (hipStream_t)stream,
                           value, spatial_shapes, level_start_index, raw, ld_raw, ref, output, batch, num_query,
                           value_batch_stride, value_row_stride, valid_ratios, 0, num_query);
    else
        hipLaunchKernelGGL((msda_fused_kernel<4, true>), dim3((unsigned)cdiv(nq, 4)), dim3(256), 0, (hipStream_t)stream, value,
                           spatial_shapes, level_start_index, raw, ld_raw, ref, output, batch, num_query,
                           value_batch_stride, value_row_stride, valid_ratios);
    return gom_launch_status();
}

extern "C" int gom_ms_deform_attn_forward(const float* value, const int64_t* spatial_shapes,
                                          const int64_t* level_start_index, const float* sampling_loc,
                                          const float* attn_weight, float* output, int batch, int spatial_size,
                                          int num_heads, int channels, int num_levels, int num_query,
                                          int num_point, void* stream) {
    GOM_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && output);
    // the shape the whole DeepSolo family uses runs here; anything else on the general kernel (msda_any.hip)
    if (!(num_heads == HEADS && channels == CH && num_levels == LEVELS && num_point == 4))
        return gom_ms_deform_attn_forward_any(GOM_DTYPE_F32, value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                              output, batch, spatial_size, num_heads, channels, num_levels, num_query,
                                              num_point, stream);
    GOM_CHECK_ARG(batch > 0 && spatial_size > 0 && num_query > 0);
    const long nq = (long)batch * num_query;
    hipLaunchKernelGGL((msda_fwd_kernel<4>), dim3((unsigned)cdiv(nq, 4)), dim3(256), 0, (hipStream_t)stream, value,
                       spatial_shapes, level_start_index, sampling_loc, attn_weight, output, batch, num_query,
                       (long)spatial_size * (HEADS * CH), HEADS * CH);
    return gom_launch_status();
}

// Same op reading the value map in place from a wider row-major buffer (e.g. one 256-column slice of
// the fused [B*S, 6*256] decoder value projection): row stride / batch stride in floats.
extern "C" int gom_ms_deform_attn_forward_strided(const float* value, long value_batch_stride, int value_row_stride,
                                                  const int64_t* spatial_shapes, const int64_t* level_start_index,
                                                  const float* sampling_loc, const float* attn_weight, float* output,
                                                  int batch, int num_query, void* stream) {
    GOM_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && output);
    GOM_CHECK_ARG(batch > 0 && num_query > 0 && value_row_stride >= HEADS * CH && (value_row_stride % 4) == 0 &&
                  (value_batch_stride % 4) == 0);
    const long nq = (long)batch * num_query;
    hipLaunchKernelGGL((msda_fwd_kernel<4>), dim3((unsigned)cdiv(nq, 4)), dim3(256), 0, (hipStream_t)stream, value,
                       spatial_shapes, level_start_index, sampling_loc, attn_weight, output, batch, num_query,
                       value_batch_stride, value_row_stride);
    return gom_launch_status();
}

extern "C" int gom_msda_prepare(const float* raw, int ld_raw, const float* ref, int ref_levels,
                                const int64_t* spatial_shapes, float* sampling_loc, float* attn_weight,
                                long num_query_total, void* stream) {
    GOM_CHECK_ARG(raw && ref && spatial_shapes && sampling_loc && attn_weight);
    GOM_CHECK_ARG(ld_raw >= HEADS * LEVELS * 4 * 3 && (ref_levels == 1 || ref_levels == LEVELS));
    if (num_query_total == 0) return GOM_OK;
    hipLaunchKernelGGL((msda_prep_kernel<4>), dim3((unsigned)cdiv(num_query_total * HEADS, 256)), dim3(256), 0,
                       (hipStream_t)stream, raw, ld_raw, ref, ref_levels, spatial_shapes, sampling_loc,
                       attn_weight, num_query_total);
    return gom_launch_status();
}
