// HBM-bound glue kernels of the GoMatching path (gfx950): pre-processing, pooling, sine position
// encodings, reference-point arithmetic.  All channels-last, 16-byte accesses where the layout allows.
#include "common.h"

namespace {

// A1: (x - mean) / std, planar [B,3,H,W] -> interleaved [B,H,W,4] (4th channel zero so the stem's
// implicit-GEMM k-tiles stay 16-byte units).  gom_lstmatcher.py:159-170.
__global__ __launch_bounds__(256) void preprocess_kernel(const float* __restrict__ x, float* __restrict__ y, long HW,
                                                         long total, float m0, float m1, float m2, float s0, float s1,
                                                         float s2) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // pixel index over B*H*W
    if (i >= total) return;
    const long b = i / HW, p = i % HW;
    const float* xb = x + b * 3 * HW + p;
    f32x4 o;
    o[0] = (xb[0] - m0) / s0;
    o[1] = (xb[HW] - m1) / s1;
    o[2] = (xb[2 * HW] - m2) / s2;
    o[3] = 0.f;
    *reinterpret_cast<f32x4*>(y + i * 4) = o;
}

// A2 (stem): max_pool2d(kernel 3, stride 2, pad 1), NHWC.
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                      int C4, int OH, int OW, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // float4 index over B*OH*OW*C4
    if (i >= total) return;
    const int c = (int)(i % C4);
    long t = i / C4;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const long b = t / OH;
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int ih = oh * 2 - 1 + dy;
        if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int iw = ow * 2 - 1 + dx;
            if ((unsigned)iw >= (unsigned)W) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((b * H + ih) * W + iw) * (long)C4 + c) * 4);
            m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
        }
    }
    *reinterpret_cast<f32x4*>(y + i * 4) = m;
}

// A3 + A5: PositionalEncoding2D (normalised sine, no padding) + level_embed, written token-major
// [HW, 256] at this level's offset of the flattened buffer.  pos_encoding.py:62-82,
// deformable_transformer.py:161-163.  dim_t[128] is supplied by the host (same pow as the reference).
__global__ __launch_bounds__(256) void pos2d_kernel(const float* __restrict__ dim_t,
                                                    const float* __restrict__ level_embed, float* __restrict__ out,
                                                    int H, int W, int Hv, int Wv, float scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // over HW*256
    if (i >= (long)H * W * 256) return;
    const int ch = (int)(i & 255);
    const long tok = i >> 8;
    const int r = (int)(tok / W), c = (int)(tok % W);
    const bool is_y = ch < 128;
    const int j = is_y ? ch : ch - 128;
    const float e = is_y ? (float)(r + 1) : (float)(c + 1);
    // padded batches: the cumulative sums stop at the valid extent (pos_encoding.py:67-72); tokens beyond it are never read
    const float last = is_y ? (float)Hv : (float)Wv;
    const float emb = (e - 0.5f) / (last + 1e-6f) * scale;
    const float a = emb / dim_t[j];
    out[i] = ((j & 1) ? cosf(a) : sinf(a)) + level_embed[ch];
}

// A9: gen_point_pos_embed (adet/modeling/model/utils.py:24-37): pts [Q,2] in [0,1] -> [Q,256] (x half, y half).
__global__ __launch_bounds__(256) void point_pos_kernel(const float* __restrict__ pts, const float* __restrict__ dim_t,
                                                        float* __restrict__ out, long Q, float scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= Q * 256) return;
    const int ch = (int)(i & 255);
    const long q = i >> 8;
    const int j = ch & 127;
    const float e = pts[q * 2 + (ch >> 7)] * scale;
    const float a = e / dim_t[j];
    out[i] = (j & 1) ? cosf(a) : sinf(a);
}

__device__ __forceinline__ float inv_sigmoid(float x) {   // adet/utils/misc.py:115-119, eps 1e-5
    x = fminf(fmaxf(x, 0.f), 1.f);
    const float x1 = fmaxf(x, 1e-5f), x2 = fmaxf(1.f - x, 1e-5f);
    return logf(x1 / x2);
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// out[q, c] = sigmoid(delta[q, c] + inverse_sigmoid(ref[q, c % 2]))   (C = 2: point refinement,
// deformable_transformer.py:484-488 and detection_transformer_wobackbone.py:211-227; C = 4: boundary)
__global__ __launch_bounds__(256) void ref_sigmoid_kernel(const float* __restrict__ delta, int ld_delta,
                                                          const float* __restrict__ ref, float* __restrict__ out,
                                                          long Q, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= Q * C) return;
    const long q = i / C;
    const int c = (int)(i % C);
    out[i] = sigmoidf(delta[q * ld_delta + c] + inv_sigmoid(ref[q * 2 + (c & 1)]));
}

// The tail of a decoder layer's reference refinement and the head of the next layer's query position as ONE launch
// (deformable_transformer.py:484-488, then :470-473 + :25-37 of the following layer): a wave per point,
//   delta = h . W3^T + b3        (the last, 256 -> 2 layer of ctrl_point_coord's MLP; h = the two hidden layers' output)
//   ref'  = sigmoid(delta + inverse_sigmoid(ref))
//   pos   = sine embedding of ref' * (sx, sy)                    (skipped when pos == nullptr: the last layer)
// instead of an N = 2 GEMM launch, ref_sigmoid_kernel and point_pos_kernel (three dispatches for 10 KB of arithmetic).
__global__ __launch_bounds__(256) void ref_update_kernel(const float* __restrict__ h, int ld_h, const float* __restrict__ W3,
                                                         const float* __restrict__ b3, const float* __restrict__ ref,
                                                         const float* __restrict__ dim_t, float sx, float sy,
                                                         float* __restrict__ new_ref, float* __restrict__ pos, long Q) {
    const int lane = threadIdx.x & 63;
    const f32x4 w0 = *(const f32x4*)(W3 + lane * 4), w1 = *(const f32x4*)(W3 + 256 + lane * 4);
    const f32x4 dt = *(const f32x4*)(dim_t + ((lane * 4) & 127));
    const float bx = b3[0], by = b3[1];
    for (long q = (long)blockIdx.x * 4 + (threadIdx.x >> 6); q < Q; q += (long)gridDim.x * 4) {
        const f32x4 v = *(const f32x4*)(h + q * ld_h + lane * 4);
        float dx = v[0] * w0[0], dy = v[0] * w1[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) {
            dx = fmaf(v[i], w0[i], dx);
            dy = fmaf(v[i], w1[i], dy);
        }
        dx = wave_sum(dx) + bx;
        dy = wave_sum(dy) + by;
        const float rx = sigmoidf(dx + inv_sigmoid(ref[q * 2])), ry = sigmoidf(dy + inv_sigmoid(ref[q * 2 + 1]));
        if (lane == 0) {
            new_ref[q * 2] = rx;
            new_ref[q * 2 + 1] = ry;
        }
        if (pos) {
            const float e = (lane < 32 ? rx * sx : ry * sy) * 6.283185307179586f;       // channels [0,128) <- x, [128,256) <- y
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = e / dt[i];
                o[i] = (i & 1) ? cosf(a) : sinf(a);
            }
            *(f32x4*)(pos + q * 256 + lane * 4) = o;
        }
    }
}

// A8: per-token proposal validity for an unpadded level pyramid (deformable_transformer.py:113-133):
// valid[s] = all of ((col+0.5)/W, (row+0.5)/H) in (0.01, 0.99).
__global__ __launch_bounds__(256) void proposal_valid_kernel(const int64_t* __restrict__ shapes,
                                                             const int64_t* __restrict__ lsi, int L,
                                                             unsigned char* __restrict__ valid, long S,
                                                             const int64_t* __restrict__ vshapes) {
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    int l = 0;
    for (int i = 1; i < L; ++i) if (s >= lsi[i]) l = i;
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const long t = s - lsi[l];
    // the grid is normalised by the VALID extent of the level (deformable_transformer.py:117-124): padded tokens land
    // beyond 0.99 and are invalid by the same test that drops the border proposals
    const float Hn = (float)(vshapes ? vshapes[2 * l] : H), Wn = (float)(vshapes ? vshapes[2 * l + 1] : W);
    const float x = ((float)(t % W) + 0.5f) / Wn, y = ((float)(t / W) + 0.5f) / Hn;
    valid[s] = (x > 0.01f && x < 0.99f && y > 0.01f && y < 0.99f) ? 1 : 0;
}

// A8: top-k gather -> proposal logit add -> sigmoid -> cubic Bernstein sampling to P reference points
// (deformable_transformer.py:99-106,183-199).  coord_raw [B,S,8] = bezier_coord_embed(output_memory).
__global__ __launch_bounds__(256) void bezier_refs_kernel(const float* __restrict__ coord_raw,
                                                          const int* __restrict__ topk,
                                                          const int64_t* __restrict__ shapes,
                                                          const int64_t* __restrict__ lsi, int L,
                                                          const float* __restrict__ bern, float* __restrict__ refs,
                                                          int B, long S, int nq, int P, int compact,
                                                          const int64_t* __restrict__ vshapes) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // over B*nq*P
    if (i >= (long)B * nq * P) return;
    const int p = (int)(i % P);
    const long bq = i / P;
    const int b = (int)(bq / nq);
    const long s = topk[bq];
    int l = 0;
    for (int k = 1; k < L; ++k) if (s >= lsi[k]) l = k;
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const long t = s - lsi[l];
    const float Hn = (float)(vshapes ? vshapes[2 * l] : H), Wn = (float)(vshapes ? vshapes[2 * l + 1] : W);
    const float gx = ((float)(t % W) + 0.5f) / Wn, gy = ((float)(t / W) + 0.5f) / Hn;
    const bool ok = gx > 0.01f && gx < 0.99f && gy > 0.01f && gy < 0.99f;
    const float px = ok ? logf(gx / (1.f - gx)) : INFINITY;
    const float py = ok ? logf(gy / (1.f - gy)) : INFINITY;
    const float* cr = coord_raw + (compact ? (size_t)bq : ((size_t)b * S + s)) * 8;
    float ox = 0.f, oy = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float cx = sigmoidf(cr[2 * k] + px), cy = sigmoidf(cr[2 * k + 1] + py);
        ox = fmaf(bern[p * 4 + k], cx, ox);
        oy = fmaf(bern[p * 4 + k], cy, oy);
    }
    refs[i * 2] = ox;
    refs[i * 2 + 1] = oy;
}

// encoder reference grid for an unpadded pyramid (deformable_transformer.py:288-300, valid ratios = 1):
// ref[s] = ((col+0.5)/W, (row+0.5)/H) following the reference's linspace / divide order.
__global__ __launch_bounds__(256) void enc_ref_kernel(const int64_t* __restrict__ shapes,
                                                      const int64_t* __restrict__ lsi, int L, float* __restrict__ ref,
                                                      long S, const int64_t* __restrict__ vshapes) {
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    int l = 0;
    for (int i = 1; i < L; ++i) if (s >= lsi[i]) l = i;
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const long t = s - lsi[l];
    if (vshapes) {                                           // / (valid_ratio * size), the reference's order (:294-295)
        const float vx = (float)vshapes[2 * l + 1] / (float)W, vy = (float)vshapes[2 * l] / (float)H;
        ref[s * 2] = ((float)(t % W) + 0.5f) / (vx * (float)W);
        ref[s * 2 + 1] = ((float)(t / W) + 0.5f) / (vy * (float)H);
        return;
    }
    ref[s * 2] = ((float)(t % W) + 0.5f) / (float)W;
    ref[s * 2 + 1] = ((float)(t / W) + 0.5f) / (float)H;
}

// padded batches: value rows of tokens outside the valid extent are zero (ms_deform_attn.py:134-135); buf [B*S, ld],
// columns [col0, col0 + ncols).
__global__ __launch_bounds__(256) void zero_padded_kernel(float* __restrict__ buf, int ld, int col0, int ncols4,
                                                          const int64_t* __restrict__ shapes,
                                                          const int64_t* __restrict__ lsi,
                                                          const int64_t* __restrict__ vshapes, int L, long S, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // over B*S*ncols4
    if (i >= total) return;
    const int c = (int)(i % ncols4);
    const long tok = i / ncols4;
    const long s = tok % S;
    int l = 0;
    for (int k = 1; k < L; ++k) if (s >= lsi[k]) l = k;
    const int W = (int)shapes[2 * l + 1];
    const long t = s - lsi[l];
    if ((t / W) >= vshapes[2 * l] || (t % W) >= vshapes[2 * l + 1])
        *reinterpret_cast<f32x4*>(buf + tok * ld + col0 + c * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ o, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    *reinterpret_cast<f32x4*>(o + i * 4) =
        *reinterpret_cast<const f32x4*>(a + i * 4) + *reinterpret_cast<const f32x4*>(b + i * 4);
}

// broadcast a [rows, D] table over the batch: out[b, r, :] = src[r, :]
__global__ __launch_bounds__(256) void bcast_rows_kernel(const float* __restrict__ src, float* __restrict__ out,
                                                         long n4, long total4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    *reinterpret_cast<f32x4*>(out + i * 4) = *reinterpret_cast<const f32x4*>(src + (i % n4) * 4);
}

__global__ __launch_bounds__(256) void bcast_rows_scalar_kernel(const float* __restrict__ src, float* __restrict__ out,
                                                                long n, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) out[i] = src[i % n];
}

// detector_postprocess (gom_lstmatcher.py:100-109): x[2i] *= sx, x[2i+1] *= sy in place
__global__ __launch_bounds__(256) void scale_xy_kernel(float* __restrict__ x, long n_pairs, float sx, float sy) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pairs) return;
    x[2 * i] *= sx;
    x[2 * i + 1] *= sy;
}

}  // namespace

#define GOM_GRID(n) dim3((unsigned)cdiv((n), 256)), dim3(256), 0, (hipStream_t)stream

extern "C" int gom_preprocess_nchw_to_nhwc4(const float* images, const float* mean3, const float* std3, float* out,
                                            int B, int H, int W, void* stream) {
    GOM_CHECK_ARG(images && mean3 && std3 && out && B > 0 && H > 0 && W > 0);
    const long total = (long)B * H * W;
    hipLaunchKernelGGL(preprocess_kernel, GOM_GRID(total), images, out, (long)H * W, total, mean3[0], mean3[1],
                       mean3[2], std3[0], std3[1], std3[2]);
    return gom_launch_status();
}

extern "C" int gom_maxpool3x3s2_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, void* stream) {
    GOM_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0);
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long total = (long)B * OH * OW * (C / 4);
    hipLaunchKernelGGL(maxpool_kernel, GOM_GRID(total), x, y, H, W, C / 4, OH, OW, total);
    return gom_launch_status();
}

extern "C" int gom_pos_encoding_2d_f32(const float* dim_t128, const float* level_embed256, float* out, int H, int W,
                                       void* stream) {
    GOM_CHECK_ARG(dim_t128 && level_embed256 && out && H > 0 && W > 0);
    hipLaunchKernelGGL(pos2d_kernel, GOM_GRID((long)H * W * 256), dim_t128, level_embed256, out, H, W, H, W,
                       6.283185307179586f);
    return gom_launch_status();
}

extern "C" int gom_pos_encoding_2d_valid_f32(const float* dim_t128, const float* level_embed256, float* out, int H, int W,
                                             int valid_h, int valid_w, void* stream) {
    GOM_CHECK_ARG(dim_t128 && level_embed256 && out && H > 0 && W > 0 && valid_h > 0 && valid_h <= H && valid_w > 0 &&
                  valid_w <= W);
    hipLaunchKernelGGL(pos2d_kernel, GOM_GRID((long)H * W * 256), dim_t128, level_embed256, out, H, W, valid_h, valid_w,
                       6.283185307179586f);
    return gom_launch_status();
}

extern "C" int gom_point_pos_embed_f32(const float* pts, const float* dim_t128, float* out, long num_points,
                                       void* stream) {
    GOM_CHECK_ARG(pts && dim_t128 && out && num_points >= 0);
    if (num_points == 0) return GOM_OK;
    hipLaunchKernelGGL(point_pos_kernel, GOM_GRID(num_points * 256), pts, dim_t128, out, num_points,
                       6.283185307179586f);
    return gom_launch_status();
}

extern "C" int gom_ref_sigmoid_f32(const float* delta, int ld_delta, const float* ref, float* out, long num_points,
                                   int C, void* stream) {
    GOM_CHECK_ARG(delta && ref && out && num_points >= 0 && (C == 2 || C == 4) && ld_delta >= C);
    if (num_points == 0) return GOM_OK;
    hipLaunchKernelGGL(ref_sigmoid_kernel, GOM_GRID(num_points * C), delta, ld_delta, ref, out, num_points, C);
    return gom_launch_status();
}

extern "C" int gom_ref_update_f32(const float* h, int ld_h, const float* W3, const float* b3, const float* ref,
                                  const float* dim_t128, float sx, float sy, float* new_ref, float* pos, long num_points,
                                  void* stream) {
    GOM_CHECK_ARG(h && W3 && b3 && ref && dim_t128 && new_ref && num_points >= 0 && ld_h >= 256 && (ld_h % 4) == 0);
    GOM_CHECK_ARG(((uintptr_t)h % 16) == 0 && ((uintptr_t)W3 % 16) == 0 && ((uintptr_t)dim_t128 % 16) == 0 &&
                  (pos == nullptr || ((uintptr_t)pos % 16) == 0));
    if (num_points == 0) return GOM_OK;
    const long blocks = (num_points + 3) / 4;
    hipLaunchKernelGGL(ref_update_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, h,
                       ld_h, W3, b3, ref, dim_t128, sx, sy, new_ref, pos, num_points);
    return gom_launch_status();
}

extern "C" int gom_proposal_valid(const int64_t* spatial_shapes, const int64_t* level_start_index, int num_levels,
                                  unsigned char* valid, long S, void* stream) {
    GOM_CHECK_ARG(spatial_shapes && level_start_index && valid && num_levels > 0 && S > 0);
    hipLaunchKernelGGL(proposal_valid_kernel, GOM_GRID(S), spatial_shapes, level_start_index, num_levels, valid, S,
                       (const int64_t*)nullptr);
    return gom_launch_status();
}

extern "C" int gom_proposal_valid_masked(const int64_t* spatial_shapes, const int64_t* level_start_index, int num_levels,
                                         const int64_t* valid_shapes, unsigned char* valid, long S, void* stream) {
    GOM_CHECK_ARG(spatial_shapes && level_start_index && valid_shapes && valid && num_levels > 0 && S > 0);
    hipLaunchKernelGGL(proposal_valid_kernel, GOM_GRID(S), spatial_shapes, level_start_index, num_levels, valid, S,
                       valid_shapes);
    return gom_launch_status();
}

extern "C" int gom_bezier_reference_points(const float* coord_raw, const int* topk_idx, const int64_t* spatial_shapes,
                                           const int64_t* level_start_index, int num_levels, const float* bernstein,
                                           float* refs, int B, long S, int num_queries, int num_points,
                                           int compact, void* stream) {
    GOM_CHECK_ARG(coord_raw && topk_idx && spatial_shapes && level_start_index && bernstein && refs);
    GOM_CHECK_ARG(B > 0 && S > 0 && num_queries > 0 && num_points > 0);
    hipLaunchKernelGGL(bezier_refs_kernel, GOM_GRID((long)B * num_queries * num_points), coord_raw, topk_idx,
                       spatial_shapes, level_start_index, num_levels, bernstein, refs, B, S, num_queries, num_points,
                       compact, (const int64_t*)nullptr);
    return gom_launch_status();
}

extern "C" int gom_bezier_reference_points_masked(const float* coord_raw, const int* topk_idx, const int64_t* spatial_shapes,
                                                  const int64_t* level_start_index, const int64_t* valid_shapes,
                                                  int num_levels, const float* bernstein, float* refs, int B, long S,
                                                  int num_queries, int num_points, int compact, void* stream) {
    GOM_CHECK_ARG(coord_raw && topk_idx && spatial_shapes && level_start_index && valid_shapes && bernstein && refs);
    GOM_CHECK_ARG(B > 0 && S > 0 && num_queries > 0 && num_points > 0);
    hipLaunchKernelGGL(bezier_refs_kernel, GOM_GRID((long)B * num_queries * num_points), coord_raw, topk_idx,
                       spatial_shapes, level_start_index, num_levels, bernstein, refs, B, S, num_queries, num_points,
                       compact, valid_shapes);
    return gom_launch_status();
}

extern "C" int gom_encoder_reference_points(const int64_t* spatial_shapes, const int64_t* level_start_index,
                                            int num_levels, float* ref, long S, void* stream) {
    GOM_CHECK_ARG(spatial_shapes && level_start_index && ref && num_levels > 0 && S > 0);
    hipLaunchKernelGGL(enc_ref_kernel, GOM_GRID(S), spatial_shapes, level_start_index, num_levels, ref, S,
                       (const int64_t*)nullptr);
    return gom_launch_status();
}

extern "C" int gom_encoder_reference_points_masked(const int64_t* spatial_shapes, const int64_t* level_start_index,
                                                   const int64_t* valid_shapes, int num_levels, float* ref, long S,
                                                   void* stream) {
    GOM_CHECK_ARG(spatial_shapes && level_start_index && valid_shapes && ref && num_levels > 0 && S > 0);
    hipLaunchKernelGGL(enc_ref_kernel, GOM_GRID(S), spatial_shapes, level_start_index, num_levels, ref, S, valid_shapes);
    return gom_launch_status();
}

extern "C" int gom_zero_padded_tokens_f32(float* buf, int ld, int col0, int ncols, const int64_t* spatial_shapes,
                                          const int64_t* level_start_index, const int64_t* valid_shapes, int num_levels,
                                          int B, long S, void* stream) {
    GOM_CHECK_ARG(buf && spatial_shapes && level_start_index && valid_shapes && num_levels > 0 && B > 0 && S > 0);
    GOM_CHECK_ARG(ncols > 0 && (ncols % 4) == 0 && (col0 % 4) == 0 && (ld % 4) == 0 && col0 + ncols <= ld);
    const long total = (long)B * S * (ncols / 4);
    hipLaunchKernelGGL(zero_padded_kernel, GOM_GRID(total), buf, ld, col0, ncols / 4, spatial_shapes, level_start_index,
                       valid_shapes, num_levels, S, total);
    return gom_launch_status();
}

// 32-bit word copy as a KERNEL: the tracker's per-match descriptor upload reads the pinned (device-mapped) staging buffer
// with this instead of an async DMA, so that its ordering against the next kernel of the stream is plain kernel order.
namespace {
__global__ __launch_bounds__(256) void copy_words_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void flag_nonfinite_kernel(const float* __restrict__ x, long n, int* __restrict__ flag) {
    bool bad = false;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) bad |= !(fabsf(x[i]) <= 3.4e38f);
    if (bad) atomicOr(flag, 1);
}
}  // namespace

extern "C" int gom_flag_nonfinite_f32(const float* x, long n, int* flag, void* stream) {
    GOM_CHECK_ARG(flag && n >= 0 && (x || n == 0));
    if (n == 0) return GOM_OK;
    const long blocks = cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048;
    hipLaunchKernelGGL(flag_nonfinite_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, flag);
    return gom_launch_status();
}

extern "C" int gom_copy_words(const void* src, void* dst, long n_words, void* stream) {
    GOM_CHECK_ARG(src && dst && n_words >= 0);
    if (n_words == 0) return GOM_OK;
    hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)cdiv(n_words, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned*)src, (unsigned*)dst, n_words);
    return gom_launch_status();
}

extern "C" int gom_add_f32(const float* a, const float* b, float* out, long n, void* stream) {
    GOM_CHECK_ARG(a && b && out && n >= 0 && (n % 4) == 0);
    if (n == 0) return GOM_OK;
    hipLaunchKernelGGL(add_kernel, GOM_GRID(n / 4), a, b, out, n / 4);
    return gom_launch_status();
}

extern "C" int gom_broadcast_rows_f32(const float* src, float* out, long n, int B, void* stream) {
    GOM_CHECK_ARG(src && out && n > 0 && B > 0);
    if (n % 4)
        hipLaunchKernelGGL(bcast_rows_scalar_kernel, GOM_GRID(n * B), src, out, n, n * B);
    else
        hipLaunchKernelGGL(bcast_rows_kernel, GOM_GRID(n / 4 * B), src, out, n / 4, n / 4 * B);
    return gom_launch_status();
}

extern "C" int gom_scale_xy_f32(float* x, long n_pairs, float sx, float sy, void* stream) {
    GOM_CHECK_ARG(n_pairs >= 0 && (x || n_pairs == 0));
    if (n_pairs == 0) return GOM_OK;
    hipLaunchKernelGGL(scale_xy_kernel, GOM_GRID(n_pairs), x, n_pairs, sx, sy);
    return gom_launch_status();
}
