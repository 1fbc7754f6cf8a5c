// Output projection + residual + LayerNorm of an attention block in ONE launch (f16x3 split, fp32-class accuracy):
//
//   Y = LayerNorm( X W^T + b + R ) * gamma + beta            X, R, Y [M, 256] fp32, W [256, 256]
//
// = `src = norm1(src + self_attn_out_proj(...))` of a DeepSolo encoder layer and the three `tgt = norm_*(tgt + out_proj(...))`
// of a decoder layer (/root/reference/third_party/adet/layers/deformable_transformer.py:258-264 encoder, :386-422 decoder
// intra / inter / cross blocks).  As two launches (GEMM with the residual in its epilogue, then LayerNorm) the pair moves
// 1 + 1 + 1 | 1 + 1 = 5 KB per token through HBM for 131 kFLOP -- 26 FLOP/B, a pure HBM stream; fused it is 3 KB.
//
// Structure (round 5; the round-2 form -- 128-row tiles at one workgroup per CU, 32x32x16, the tile kernel's bits -- was one
// workgroup's 27 us latency chain whatever the launch's length and left in round 6: docs/LAB_NOTES.md): a workgroup owns 64 rows,
// a wave 16 of them as v_mfma_f32_16x16x32_f16 operand fragments, TWO workgroups per CU; see proj_ln2_kernel below.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int D = 256;                                   // model width = K = N (fixed: every shipped config)
constexpr int FRAG = 1024;                               // bytes of one MFMA operand fragment

struct ProjArgs {
    const float* X;
    const unsigned char* img;
    const float* inv;                                        // [256] 1 / row scale of the split weight
    const float* bias;
    const float* R;
    const float* gamma;
    const float* beta;
    float* Y;
    int* flag;
    float eps;
    int ldx, ldr, ldy, M;
    const float* dot_w;                                      // dot form: out[m] = <Y[m, :], dot_w> + dot_b, Y itself is not stored
    float* dot_out;
    float dot_b;
};

__device__ __forceinline__ float row16_sum(float v) {        // sum over the 16 lanes of a DPP row, result in every lane
    auto dpp = [](float x, int ctrl_tag) {
        const int xi = __builtin_bit_cast(int, x);
        int r;
        if (ctrl_tag == 0) r = __builtin_amdgcn_update_dpp(0, xi, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
        else if (ctrl_tag == 1) r = __builtin_amdgcn_update_dpp(0, xi, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
        else if (ctrl_tag == 2) r = __builtin_amdgcn_update_dpp(0, xi, 0x141, 0xF, 0xF, true);  // row_half_mirror
        else r = __builtin_amdgcn_update_dpp(0, xi, 0x140, 0xF, 0xF, true);                     // row_mirror
        return __builtin_bit_cast(float, r);
    };
    v += dpp(v, 0);
    v += dpp(v, 1);
    v += dpp(v, 2);
    v += dpp(v, 3);
    return v;
}

__device__ __forceinline__ void split2(float x, float y, unsigned int& q0, unsigned int& q1) { gom_split2_f16(x, y, q0, q1); }

__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, half8& p0, half8& p1) {
    unsigned int l0, l1, l2, l3, h0, h1, h2, h3;
    split2(a[0], a[1], l0, h0);
    split2(a[2], a[3], l1, h1);
    split2(b[0], b[1], l2, h2);
    split2(b[2], b[3], l3, h3);
    p0 = __builtin_bit_cast(half8, (u32x4{l0, l1, l2, l3}));
    p1 = __builtin_bit_cast(half8, (u32x4{h0, h1, h2, h3}));
}

__device__ __forceinline__ void dma_fragment(__amdgpu_buffer_rsrc_t rs, unsigned byte_offset, unsigned char* lds_frag) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_frag, 16, (int)byte_offset, 0, 0, 0);
}

// FORM 0: + residual, rows stored (the attention blocks); 1: no residual, rows stored; 2: no residual, only <row, dot_w> + dot_b
// stored.  Compile-time: as run-time branches the residual loads of form 0 were no longer batched (268 -> 405 us at 297k rows).
// ---- round 5: the TWO-WORKGROUPS-PER-CU form ----
// The round-2 kernel was one workgroup's serial chain -- rows in (HBM latency), 384 MFMAs per wave, staged epilogue with the residual
// in and the rows out -- at one workgroup per CU: a 128-row tile took ~27 us whether the launch had 157 tiles or 2 324, i.e. the
// launch was bound by that chain's latency, not by HBM (3.6 of 8 TB/s at M = 297 368).  Here a workgroup owns 64 rows -- a wave 16,
// as v_mfma_f32_16x16x32_f16 operand fragments in 64 VGPRs, sixteen 16 x 16 accumulators in 64 more -- so that its <= 256 registers
// and 64 KB of LDS (a two-stage ring of 32 KB k-step stages = the epilogue's staging area) let TWO workgroups share a CU: one's
// row / residual / store phases run under the other's MFMAs, and twice as many loads are in flight per CU.  The weights cross
// L2 -> LDS twice as often per row (256 KB per 64 rows); the loop's LDS reads per MFMA double too (a fragment serves 16 rows) and
// stay at half the LDS rate.  Arithmetic: the fused FFN kernel's second product (k-steps of 32, plane products lo-hi, hi-lo,
// hi-hi): fp32-class like the form above, not its bits; the three FORMs agree with each other bit for bit.
constexpr int V2_BM = 64;
constexpr int V2_STAGE_FRAGS = (D / 16) * 2;             // one 32-wide k-step: 16 output groups x 2 planes
constexpr int V2_STAGE_BYTES = V2_STAGE_FRAGS * FRAG;    // 32 KB
constexpr int V2_STAGES = D / 32;                        // 8
constexpr int V2_LDS_BYTES = 2 * V2_STAGE_BYTES;         // = 64 rows x 256 fp32: the staging area fits the ring exactly
constexpr long V2_IMAGE_BYTES = (long)V2_STAGES * V2_STAGE_BYTES;

__device__ __forceinline__ f32x4 mfma16(const half8 a, const half8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <int FORM>
__global__ __launch_bounds__(256, 2) void proj_ln2_kernel(const ProjArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fn = lane & 15, fg = lane >> 4;
    const long tile0 = (long)blockIdx.x * V2_BM;
    const long row0 = tile0 + wave * 16;
    const unsigned char* img = p.img;
    const __amdgpu_buffer_rsrc_t rs_img = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, (int)V2_IMAGE_BYTES, 0x00020000);

    // ---- this wave's 16 rows as B-operand fragments: lane (n, kg) holds X[row n][32 s + 8 kg .. + 7], two planes.  Whole-line loads
    // (a wave-instruction = 128 floats of 2 rows) + a layout change in a wave-private 8 KB of the ring's second slot ----
    int range_bad = 0;
    half8 xf[2][D / 32];
    {
        float xmax = 0.f;
        float* scratch = reinterpret_cast<float*>(smem + V2_STAGE_BYTES) + wave * (16 * 128);
        const int pc = lane & 31, r0 = lane >> 5;                // 16-byte piece of a 128-float part, row inside a pair
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            f32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                long m = row0 + r0 + 2 * i;
                if (m > p.M - 1) m = p.M - 1;                 // tail rows recompute the last row (never stored)
                v[i] = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * p.ldx + part * 128 + pc * 4);
            }
            if (part == 0) {
                __builtin_amdgcn_sched_barrier(0);
                for (int f = wave; f < V2_STAGE_FRAGS; f += 4) dma_fragment(rs_img, f * FRAG + lane * 16, smem + f * FRAG);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = r0 + 2 * i;
                *reinterpret_cast<f32x4*>(scratch + r * 128 + ((pc ^ (r & 31)) << 2)) = v[i];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                const int p0 = 8 * s_ + 2 * fg;
                const f32x4 a = *reinterpret_cast<const f32x4*>(scratch + fn * 128 + ((p0 ^ (fn & 31)) << 2));
                const f32x4 b = *reinterpret_cast<const f32x4*>(scratch + fn * 128 + (((p0 + 1) ^ (fn & 31)) << 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) xmax = fmaxf(xmax, fmaxf(fabsf(a[e]), fabsf(b[e])));
                split8(a, b, xf[0][4 * part + s_], xf[1][4 * part + s_]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        asm volatile("" : "+v"(xmax));
        range_bad = !(xmax <= 65504.f);
    }

    f32x4 acc[D / 16];
#pragma unroll
    for (int t = 0; t < D / 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    constexpr unsigned OOB = 0x7FFF0000u;
#pragma unroll
    for (int c = 0; c < V2_STAGES; ++c) {
        const int st = c & 1;
        const unsigned nsrc = c + 1 < V2_STAGES ? (unsigned)(c + 1) * V2_STAGE_BYTES + wave * FRAG + lane * 16 : OOB;
        unsigned char* ndst = smem + (st ^ 1) * V2_STAGE_BYTES + wave * FRAG;
        const unsigned char* base = smem + st * V2_STAGE_BYTES + lane * 16;
        half8 fa[8], fb[8];
#define P2_DMA(i) dma_fragment(rs_img, nsrc + (i) * 4 * FRAG, ndst + (i) * 4 * FRAG);
#define P2_LOAD(dst, g)                                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                          \
        dst[i_] = *reinterpret_cast<const half8*>(base + ((g) * 8 + i_) * FRAG);
#define P2_PIN()                                          \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        // group g of a stage: output groups 4 g .. 4 g + 3; fragment 2 i + p = plane p of output group 4 g + i
#define P2_MFMA(src, g)                                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                        \
        const int t_ = (g) * 4 + i_;                                                                          \
        acc[t_] = mfma16(src[2 * i_ + 1], xf[0][c], acc[t_]);                                                 \
        acc[t_] = mfma16(src[2 * i_], xf[1][c], acc[t_]);                                                     \
        acc[t_] = mfma16(src[2 * i_], xf[0][c], acc[t_]);                                                     \
    }
        P2_LOAD(fa, 0)
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        P2_LOAD(fb, 1) P2_MFMA(fa, 0) P2_DMA(0) P2_DMA(1) P2_PIN()
        P2_LOAD(fa, 2) P2_MFMA(fb, 1) P2_DMA(2) P2_DMA(3) P2_PIN()
        P2_LOAD(fb, 3) P2_MFMA(fa, 2) P2_DMA(4) P2_DMA(5) P2_PIN()
        P2_MFMA(fb, 3) P2_DMA(6) P2_DMA(7)
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
#undef P2_DMA
#undef P2_LOAD
#undef P2_PIN
#undef P2_MFMA
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: Y^T (row on the lane, features 16 t + 4 g .. + 3 in the registers) -> row-major through the ring ----
    float* stg = reinterpret_cast<float*>(smem);             // [64][256] fp32; 16-byte chunk c of row r at chunk c ^ (r & 7)
    {
        const int lr = wave * 16 + fn;
        float* mine = stg + lr * D;
#pragma unroll
        for (int t = 0; t < D / 16; ++t) {
            const int chunk = 4 * t + fg;
            *reinterpret_cast<f32x4*>(mine + ((chunk ^ (lr & 7)) << 2)) = acc[t];
        }
    }
    const int sub = lane & 15, rsel = lane >> 4;
    // the residual rows of the row pass are all requested here, under the barrier and the first staged reads
    f32x4 xres[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const long m = tile0 + wave * 16 + 4 * g + rsel;
        const long mc = m < p.M ? m : p.M - 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (FORM == 0) xres[g][k] = *reinterpret_cast<const f32x4*>(p.R + (size_t)mc * p.ldr + (sub + 16 * k) * 4);
            else xres[g][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
    f32x4 sc[4], bi[4], ga[4], be[4], dw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int col = (sub + 16 * k) * 4;
        sc[k] = *reinterpret_cast<const f32x4*>(p.inv + col);
        bi[k] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        ga[k] = *reinterpret_cast<const f32x4*>(p.gamma + col);
        be[k] = *reinterpret_cast<const f32x4*>(p.beta + col);
        dw[k] = FORM == 2 ? *reinterpret_cast<const f32x4*>(p.dot_w + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int bad = range_bad;
    f32x4 v[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int lr = wave * 16 + 4 * g + rsel;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ch = sub + 16 * k;
            v[g][k] = *reinterpret_cast<const f32x4*>(stg + lr * D + ((ch ^ (lr & 7)) << 2));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const long m = tile0 + wave * 16 + 4 * g + rsel;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (FORM == 0) v[g][k] = v[g][k] * sc[k] + bi[k] + xres[g][k];
            else v[g][k] = v[g][k] * sc[k] + bi[k];
            sum += (v[g][k][0] + v[g][k][1]) + (v[g][k][2] + v[g][k][3]);
        }
        const float mean = row16_sum(sum) * (1.f / D);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[g][k] = v[g][k] - mean;
            q += (v[g][k][0] * v[g][k][0] + v[g][k][1] * v[g][k][1]) + (v[g][k][2] * v[g][k][2] + v[g][k][3] * v[g][k][3]);
        }
        const float rstd = rsqrtf(row16_sum(q) * (1.f / D) + p.eps);
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 o = v[g][k] * rstd * ga[k] + be[k];
            bad |= !(fabsf(o[0]) <= 3.4e38f) | !(fabsf(o[1]) <= 3.4e38f) | !(fabsf(o[2]) <= 3.4e38f) | !(fabsf(o[3]) <= 3.4e38f);
            if constexpr (FORM == 2) dot += (o[0] * dw[k][0] + o[1] * dw[k][1]) + (o[2] * dw[k][2] + o[3] * dw[k][3]);
            else if (m < p.M) *reinterpret_cast<f32x4*>(p.Y + (size_t)m * p.ldy + (sub + 16 * k) * 4) = o;
        }
        if constexpr (FORM == 2) {
            dot = row16_sum(dot) + p.dot_b;
            if (sub == 0 && m < p.M) p.dot_out[m] = dot;
        }
    }
    if (bad && p.flag) atomicOr(p.flag, 1);
}

// the image: stage c = k-step c (32 inputs); fragment f = 2 t + p: element j of lane (m, kg) = plane p of
// Ws[16 t + m][32 c + 8 kg + j].
__global__ __launch_bounds__(256) void proj_ln2_image_kernel(const unsigned short* __restrict__ planes, long plane_stride, int ldw,
                                                             unsigned short* __restrict__ img) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)V2_STAGES * V2_STAGE_FRAGS * 512;
    if (idx >= total) return;
    const int e = (int)(idx % 512), f = (int)((idx / 512) % V2_STAGE_FRAGS), c = (int)(idx / (512L * V2_STAGE_FRAGS));
    const int l = e >> 3, j = e & 7, m = l & 15, kg = l >> 4;
    const int t = f >> 1, pl = f & 1;
    img[idx] = planes[pl * plane_stride + (size_t)(16 * t + m) * ldw + 32 * c + 8 * kg + j];
}

}  // namespace

extern "C" long gom_proj_ln_image_bytes(int n, int k) {
    if (n != D || k != D) return -1;
    return V2_IMAGE_BYTES;
}

extern "C" int gom_proj_ln_image(const void* w_planes, long w_plane_stride, int ldw, int n, int k, void* image,
                                 long image_bytes, void* stream) {
    GOM_CHECK_ARG(w_planes && image && n == D && k == D && ldw >= D);
    GOM_CHECK_ARG(image_bytes >= gom_proj_ln_image_bytes(n, k));
    const long total2 = (long)V2_STAGES * V2_STAGE_FRAGS * 512;
    hipLaunchKernelGGL(proj_ln2_image_kernel, dim3((unsigned)cdiv(total2, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)w_planes, w_plane_stride, ldw, (unsigned short*)image);
    return gom_launch_status();
}

template <int FORM>
static int proj_ln_launch_t(const ProjArgs& a, hipStream_t stream) {
    // (the attribute is per DEVICE: set on every launch -- a process-wide flag would miss a second GPU; it costs ~1 us)
    hipError_t e2 = hipFuncSetAttribute((const void*)proj_ln2_kernel<FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS_BYTES);
    if (e2 != hipSuccess) return GOM_ERR_HIP_BASE + (int)e2;
    hipLaunchKernelGGL(proj_ln2_kernel<FORM>, dim3((unsigned)cdiv(a.M, V2_BM)), dim3(256), V2_LDS_BYTES, stream, a);
    return gom_launch_status();
}

static int proj_ln_launch(ProjArgs a, hipStream_t stream) {
    if (a.dot_out) return proj_ln_launch_t<2>(a, stream);
    return a.R ? proj_ln_launch_t<0>(a, stream) : proj_ln_launch_t<1>(a, stream);
}

// R may be NULL: Y = LayerNorm(X W^T + b) (the encoder's enc_output + enc_output_norm pair, deformable_transformer.py:171-172)
extern "C" int gom_proj_ln_f32(const float* X, int ldx, const void* image, const float* w_inv_scale, const float* bias,
                               const float* R, int ldr, const float* gamma, const float* beta, float eps, float* Y, int ldy,
                               int M, int* flag, void* stream) {
    GOM_CHECK_ARG(X && image && w_inv_scale && gamma && beta && Y && M >= 0);
    GOM_CHECK_ARG(ldx >= D && (!R || ldr >= D) && ldy >= D && (ldx % 4) == 0 && (ldr % 4) == 0 && (ldy % 4) == 0);
    GOM_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)R % 16) == 0 && ((uintptr_t)Y % 16) == 0 &&
                  ((uintptr_t)image % 16) == 0 && ((uintptr_t)w_inv_scale % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0) &&
                  ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0);
    if (M == 0) return GOM_OK;
    ProjArgs a{};
    a.X = X; a.img = (const unsigned char*)image; a.inv = w_inv_scale; a.bias = bias; a.R = R; a.gamma = gamma; a.beta = beta;
    a.Y = Y; a.flag = flag; a.eps = eps; a.ldx = ldx; a.ldr = ldr; a.ldy = ldy; a.M = M;
    return proj_ln_launch(a, (hipStream_t)stream);
}

// out[m] = < LayerNorm(X[m] W^T + b) * gamma + beta , dot_w > + dot_b: the encoder's proposal class logit of EVERY token
// (deformable_transformer.py:171-175 enc_output -> enc_output_norm -> bezier_class_embed) without the normalised rows ever
// reaching HBM -- the 100 winners per frame are recomputed by gom_proj_ln_f32 on their gathered rows (same bits, row by row).
extern "C" int gom_proj_ln_dot_f32(const float* X, int ldx, const void* image, const float* w_inv_scale, const float* bias,
                                   const float* gamma, const float* beta, float eps, const float* dot_w, float dot_b,
                                   float* out, int M, int* flag, void* stream) {
    GOM_CHECK_ARG(X && image && w_inv_scale && gamma && beta && dot_w && out && M >= 0);
    GOM_CHECK_ARG(ldx >= D && (ldx % 4) == 0);
    GOM_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)image % 16) == 0 && ((uintptr_t)w_inv_scale % 16) == 0 &&
                  (!bias || ((uintptr_t)bias % 16) == 0) && ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0 &&
                  ((uintptr_t)dot_w % 16) == 0);
    if (M == 0) return GOM_OK;
    ProjArgs a{};
    a.X = X; a.img = (const unsigned char*)image; a.inv = w_inv_scale; a.bias = bias; a.R = nullptr; a.gamma = gamma;
    a.beta = beta; a.Y = nullptr; a.flag = flag; a.eps = eps; a.ldx = ldx; a.M = M;
    a.dot_w = dot_w; a.dot_b = dot_b; a.dot_out = out;
    return proj_ln_launch(a, (hipStream_t)stream);
}
