// Row-resident K = 256 GEMM on the fp16 matrix cores (f16x3 split: gemm_f16x3.hip) for the SHORT problems of the DeepSolo
// decoder:   C[M, N] = act( (A [+ A2])[M, 256] . W[N, 256]^T / rowscale + bias  [+ R] )
//
// = every Q-side nn.Linear of a composite decoder layer (/root/reference/third_party/adet/layers/deformable_transformer.py
//   :386-422 intra / inter attention in- and out-projections, cross-attention offsets|logits and output projection, :470-488
//   reference-point MLPs) at M = frames x queries x points = 20 000 rows.  The 128x128 tile kernel is LATENCY-bound there: 157 x
//   N/128 workgroups never fill the chip and one launch lasts as long as ONE tile's serial chain (8 k-steps, each global load ->
//   split -> LDS -> barrier -> 24 MFMAs -> barrier; 23-48 us for 2.6 us of MFMA work, profiles/r02_gemm_shapes.csv).
// Here the chain is: ONE round of loads (the wave's 32 rows of A, whole K, split once into MFMA operand fragments that stay in
// 128 VGPRs -- the fused FFN kernel's scheme, ffn_fused.hip), then per 32 output columns 48 MFMAs whose weight fragments stream
// through a two-stage LDS ring by LDS-DMA from a fragment-linear image (no VALU, no bank conflicts, one barrier per 48 MFMAs),
// and the accumulator goes straight to global memory (row of A = lane, 4 consecutive columns per register quad: 16-byte
// stores).  blockIdx.y can split the columns over several workgroups (col_groups; <= 0 = ~2 workgroups per CU): every group
// then re-splits its rows, and in the step one group measured faster (689 vs 878 us for the decoder's 30 launches) -- the
// product launches with col_groups = 1, tests force the others.
// The plane products run in the tile kernel's order (A-lo x W-hi, A-hi x W-lo, A-hi x W-hi per 16-wide k-step, k ascending),
// and the epilogue is the same fma: results are bit-identical to gom_gemm_f32_f16x3 (tests/test_gemm_k256_gpu.py).
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int KD = 256;                                  // K (fixed: d_model of every shipped config)
constexpr int CW = 32;                                   // output columns per chunk
constexpr int FRAG = 1024;                               // bytes of one MFMA operand fragment
constexpr int W_FRAGS = (KD / 16) * 2;                   // k-steps x planes
constexpr int CHUNK_FRAGS = W_FRAGS + 1;                 // + (1 / row scale | bias) of the chunk's 32 columns
constexpr int CHUNK_BYTES = CHUNK_FRAGS * FRAG;
constexpr int BM = 128;
constexpr int LDS_BYTES = 2 * CHUNK_BYTES;

struct RowArgs {
    const float* A;
    const float* A2;
    const unsigned char* img;
    const float* R;
    float* C;
    int* flag;
    int lda, ldr, ldc, M, chunks, cpg, r_chunks, relu, r_period;
    int tiles, frames, tpf;                                  // frame-interleaved tile order (periodic residual): see tile_of()
};

// Which 128-row tile a workgroup takes.  Plain launches: its index.  With a PERIODIC residual (the encoder's position table
// [S, 384]: row m reads table row m % S) the same 196 KB of table are needed by the tiles at the same position of every frame,
// and in index order those are a whole frame (S rows, 57 MB of table) apart: every frame re-reads the table through the
// fabric (profiles/r03: 1.30 x the algorithmic bytes).  Workgroups go round-robin to the 8 XCDs by their linear index, each XCD
// with its own 4 MB L2 -- so XCD x = i % 8 takes, in the order j = i / 8 it starts them, the `frames` tiles of position
// q = (j / frames) * 8 + x one after the other: the table rows of a position are fetched once per XCD pass and hit in L2 for the
// other frames.  Tiles straddle frame boundaries (S % 128 != 0), which only shifts a position's rows by < 128: still the same
// lines.  Ids beyond the last position / tile exit at once (the id space is padded to whole groups of 8 x frames).
// MEASURED (tools/gemm_k256_bench.py, M = 297 368, N = 640): 374 us against 364 us in index order -- the re-reads are served by
// the 256 MB Infinity Cache at no visible cost, and the interleave breaks the write stream's order.  Kept as an option
// (gom_gemm_k256_set_interleave), OFF by default.
__device__ __forceinline__ long tile_of(const RowArgs& p, unsigned i) {
    if (p.frames <= 1) return i;
    const unsigned x = i & 7u, j = i >> 3;
    const unsigned b = j % (unsigned)p.frames, q = (j / (unsigned)p.frames) * 8u + x;
    if (q >= (unsigned)p.tpf) return -1;
    const long t = (long)b * p.tpf + q;
    return t < p.tiles ? t : -1;
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

template <bool ROWS_FIRST>
__device__ __forceinline__ f32x16 mfma_either(const half8 w, const half8 x, const f32x16 acc) {
    if constexpr (ROWS_FIRST) return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, w, acc, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, acc, 0, 0, 0);
}

// MUBUF LDS-DMA (not global_load_lds: see ffn_fused.hip -- the FLAT form turns every counted lgkmcnt wait into lgkmcnt(0))
__device__ __forceinline__ void dma_fragment(__amdgpu_buffer_rsrc_t rs, unsigned byte_offset, unsigned char* lds_frag) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_frag, 16, (int)byte_offset, 0, 0, 0);
}

// LINES = false: C^T chunk = Wc . A^T, the lane is the ROW (four 16-byte stores per chunk, each 32-byte pieces of 32 lines);
// LINES = true: C chunk = A . Wc^T, the lane is the COLUMN and register r holds row (r & 3) + 8 (r >> 2) + 4 h of the wave's 32
// (sixteen 4-byte stores per chunk, each two WHOLE 128-byte lines).  Same products, same order: the same bits (tests).  Long
// problems are bound by their store path -- with the stores removed the kernel takes 25 % less time (tools/exp/k256_clock.py:
// 368 -> 274 us at M = 297 368, N = 640) -- and whole lines cost it 6-7 % less (368 -> 345 us; N = 1536: 816 -> 763 us); short
// ones (the decoder, less than one workgroup per CU) are not, and pay 3 % for the twelve extra store instructions (18.7 -> 19.5 us
// at M = 20 000, N = 256): the launcher picks by M.
template <bool LINES>
__global__ __launch_bounds__(256, 2) void gemm_k256_kernel(const RowArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int c0 = (int)blockIdx.y * p.cpg;
    const int c1 = min(p.chunks, c0 + p.cpg);
    const long tile = tile_of(p, blockIdx.x);
    if (tile < 0) return;                                    // padding of the interleaved id space (whole workgroup)
    long row = tile * BM + wave * 32 + fr;
    if (row > p.M - 1) row = p.M - 1;                        // tail rows recompute AND re-store the last row (same bits)
    const __amdgpu_buffer_rsrc_t rs_img =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.img, 0, p.chunks * CHUNK_BYTES, 0x00020000);
    auto dma_stage = [&](int c, int stage) {                 // 33 fragments, dealt to the four waves
        const unsigned src = (unsigned)c * CHUNK_BYTES + lane * 16;
        unsigned char* dst = smem + stage * CHUNK_BYTES;
        for (int f = wave; f < CHUNK_FRAGS; f += 4) dma_fragment(rs_img, src + f * FRAG, dst + f * FRAG);
    };
    constexpr unsigned OOB = 0x7FFF0000u;                    // beyond num_records: the DMA writes zeros (into an unused stage)

    // ---- this wave's 32 rows as operand fragments: lane (r, h) holds A[row r][16 s + 8 h .. + 7], two planes ----
    // (whole-line loads + a layout change in the ring's second slot, 64 of a row's floats at a time; the first stage is requested
    // behind the first loads: common.h gom_rows_to_fragments)
    int bad = 0;
    half8 xf[2][KD / 16];
    {
        float xmax = 0.f;
        const long wrow0 = tile * BM + wave * 32;
        auto arow = [&](int r) {
            long m = wrow0 + r;
            if (m > p.M - 1) m = p.M - 1;                     // tail rows recompute (and re-store) the last row: same bits
            return p.A + (size_t)m * p.lda;
        };
        auto a2row = [&](int r) {
            long m = wrow0 + r;
            if (m > p.M - 1) m = p.M - 1;
            return p.A2 + (size_t)m * p.lda;
        };
        float* scratch = reinterpret_cast<float*>(smem + CHUNK_BYTES) + wave * (32 * 64);
        if (p.A2) gom_rows_to_fragments<64, true>(arow, a2row, scratch, lane, xf, xmax, [&]() { dma_stage(c0, 0); });
        else gom_rows_to_fragments<64, false>(arow, arow, scratch, lane, xf, xmax, [&]() { dma_stage(c0, 0); });
        bad = !(xmax <= 65504.f);
    }

    f32x16 acc;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // one chunk: 64 weight fragments in eight-fragment groups, group g + 1 read while the twelve MFMAs of group g issue
    // (explicit two-deep pipeline pinned by sched_group_barrier, as in ffn_fused.hip)
    // `nsrc` / `ndst`: this wave's eight weight fragments of the NEXT stage (fragments wave, wave + 4, ...), issued one per
    // four MFMAs: an LDS-DMA instruction costs its wave 100-140 cycles of issue (measured with s_memtime stamps: 1250 of the
    // 4200 cycles of a chunk when all nine were issued in front of the MFMAs); beside running MFMAs that time is hidden.
    auto chunk_product = [&](const unsigned char* base, f32x16& acc, unsigned nsrc, unsigned char* ndst) {
        half8 fa[8], fb[8];
#define K256_LOAD(dst, g)                                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                          \
        dst[i_] = *reinterpret_cast<const half8*>(base + ((g) * 8 + i_) * FRAG);
#define K256_DMA(i) dma_fragment(rs_img, nsrc + (i) * 4 * FRAG, ndst + (i) * 4 * FRAG);
#define K256_PIN3()                                       \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
#define K256_PIN2()                                       \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        // C^T chunk [32 columns x 32 rows] = Wc . A^T: A operand = weight fragment (LDS), B operand = the rows in registers
#define K256_MFMA(src, g)                                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                        \
        const int s_ = (g) * 4 + i_;                                                                          \
        acc = mfma_either<LINES>(src[2 * i_], xf[1][s_], acc);                                                \
        acc = mfma_either<LINES>(src[2 * i_ + 1], xf[0][s_], acc);                                            \
        acc = mfma_either<LINES>(src[2 * i_], xf[0][s_], acc);                                                \
    }
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.f;
        K256_LOAD(fa, 0)
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        K256_LOAD(fb, 1) K256_MFMA(fa, 0) K256_DMA(0) K256_DMA(1) K256_DMA(2) K256_PIN3()
        K256_LOAD(fa, 2) K256_MFMA(fb, 1) K256_DMA(3) K256_DMA(4) K256_DMA(5) K256_PIN3()
        K256_LOAD(fb, 3) K256_MFMA(fa, 2) K256_DMA(6) K256_DMA(7) K256_PIN2()
        K256_MFMA(fb, 3)
#undef K256_LOAD
#undef K256_DMA
#undef K256_PIN3
#undef K256_PIN2
#undef K256_MFMA
    };

    {
        const float lo = p.relu ? 0.f : -INFINITY;
        // residual row of this lane (LINES = false); `r_period` > 0: row m reads R[m % r_period] (a table repeated per frame)
        const long rrow = p.r_period > 0 ? row % p.r_period : row;
        // LINES: output (and residual) rows of this workgroup through buffer descriptors that end with its last valid row --
        // stores to rows >= M are dropped by the bounds check and still ISSUED, so that the counted wait below stays exact.
        // Lane (column fr, half h) addresses row 4 h of its wave's 32; register r adds (r & 3) + 8 (r >> 2) rows (in the
        // VECTOR offset: the bounds check does not see the scalar one).
        const long tile0 = tile * BM;
        const unsigned rows_here = (unsigned)min((long)BM, (long)p.M - tile0);
        const unsigned c_row = (unsigned)p.ldc * 4u, r_row = (unsigned)p.ldr * 4u;
        const __amdgpu_buffer_rsrc_t rs_c =
            __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + (size_t)tile0 * p.ldc), 0, (int)(rows_here * c_row), 0x00020000);
        const bool periodic = p.r_period > 0;
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(!p.R ? p.C : periodic ? p.R : p.R + (size_t)tile0 * p.ldr), 0,
            !p.R ? 0 : (int)((periodic ? (unsigned)p.r_period : rows_here) * r_row), 0x00020000);
        const unsigned c_lane = (unsigned)(wave * 32 + 4 * fh) * c_row;
        const unsigned r_first = periodic ? (unsigned)((tile0 + wave * 32 + 4 * fh) % p.r_period) : (unsigned)(wave * 32 + 4 * fh);
        auto r_offset = [&](int k) {                         // byte offset of residual row (first + k), wrapped once (period >= 32)
            unsigned rr = r_first + (unsigned)k;
            if (periodic && rr >= (unsigned)p.r_period) rr -= (unsigned)p.r_period;
            return rr * r_row;
        };
        for (int c = c0; c < c1; ++c) {
            const int st = (c - c0) & 1;
            const bool more = c + 1 < c1;
            if (more && wave == 0)                           // the (scale | bias) fragment of the next stage
                dma_fragment(rs_img, (unsigned)(c + 1) * CHUNK_BYTES + W_FRAGS * FRAG + lane * 16,
                             smem + (st ^ 1) * CHUNK_BYTES + W_FRAGS * FRAG);
            const unsigned nsrc = more ? (unsigned)(c + 1) * CHUNK_BYTES + wave * FRAG + lane * 16 : OOB;
            unsigned char* ndst = smem + (st ^ 1) * CHUNK_BYTES + wave * FRAG;
            const bool use_r = p.R && c < p.r_chunks;
            const float* aux = reinterpret_cast<const float*>(smem + st * CHUNK_BYTES + W_FRAGS * FRAG);
            if constexpr (!LINES) {
                f32x4 rv[4];                                 // residual quads of this chunk: issued before the MFMAs
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    rv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (use_r) rv[q] = *reinterpret_cast<const f32x4*>(p.R + (size_t)rrow * p.ldr + CW * c + 8 * q + 4 * fh);
                }
                chunk_product(smem + st * CHUNK_BYTES + lane * 16, acc, nsrc, ndst);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(aux + 8 * q + 4 * fh);
                    const f32x4 bi = *reinterpret_cast<const f32x4*>(aux + CW + 8 * q + 4 * fh);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = fmaf(acc[4 * q + e], sc[e], bi[e]) + rv[q][e];
                        bad |= !(fabsf(t) <= 3.4e38f);         // before the ReLU: fmaxf(NaN, 0) = 0 would hide it
                        v[e] = fmaxf(t, lo);
                    }
                    *reinterpret_cast<f32x4*>(p.C + (size_t)row * p.ldc + CW * c + 8 * q + 4 * fh) = v;
                }
                // the four stores may stay in flight; everything older (the next stage's DMA, the residual loads) has landed
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                const unsigned voff = (unsigned)(CW * c + fr) * 4u;
                chunk_product(smem + st * CHUNK_BYTES + lane * 16, acc, nsrc, ndst);
                const float sc = aux[fr], bi = aux[CW + fr];
                // (the residual is read HERE, not in front of the MFMAs: sixteen more live registers there spill -- two
                // workgroups per CU leave 256 -- and the other workgroup's MFMAs run under this latency)
                float rv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    rv[r] = 0.f;
                    if (use_r) rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, (int)(voff + r_offset((r & 3) + 8 * (r >> 2))), 0, 0));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float t = fmaf(acc[r], sc, bi) + rv[r];
                    bad |= !(fabsf(t) <= 3.4e38f);             // before the ReLU: fmaxf(NaN, 0) = 0 would hide it
                    const float v = fmaxf(t, lo);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_c, (int)(voff + c_lane + ((r & 3) + 8 * (r >> 2)) * c_row), 0, 0);
                }
                // the sixteen stores may stay in flight; everything older has landed
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            }
            __syncthreads();                                 // next stage complete for everybody; nobody still reads this one
        }
    }

    if (bad && p.flag) atomicOr(p.flag, 1);                  // an operand left fp16's range (gemm_f16x3.hip contract)
}

// Fragment-linear image of W[N, 256] (row-scaled planes of gom_split_f16x2).  Per chunk c of 32 output columns 33 KB:
//   f = 2 s + p (s = 0..15): element j of lane (r, h) = plane p of Ws[32 c + r][16 s + 8 h + j]
//   f = 32: floats 0..31 = 1 / row scale, 32..63 = bias of the chunk's columns
__global__ __launch_bounds__(256) void k256_image_kernel(const unsigned short* __restrict__ planes, long plane_stride, int ldw,
                                                         const float* __restrict__ inv, const float* __restrict__ bias, int N,
                                                         unsigned short* __restrict__ img) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)(N / CW) * CHUNK_FRAGS * 512;
    if (i >= total) return;
    const int e = (int)(i % 512), f = (int)((i / 512) % CHUNK_FRAGS), c = (int)(i / (512L * CHUNK_FRAGS));
    const int l = e >> 3, j = e & 7, r = l & 31, h = l >> 5;
    if (f < W_FRAGS) {
        const int s = f >> 1, pl = f & 1;
        img[i] = planes[pl * plane_stride + (size_t)(CW * c + r) * ldw + 16 * s + 8 * h + j];
    } else {
        const int fi = e >> 1;
        float v = 0.f;
        if (fi < CW) v = inv[CW * c + fi];
        else if (fi < 2 * CW) v = bias ? bias[CW * c + fi - CW] : 0.f;
        const unsigned bits = __builtin_bit_cast(unsigned, v);
        img[i] = (unsigned short)((e & 1) ? (bits >> 16) : (bits & 0xffffu));
    }
}

}  // namespace

extern "C" long gom_gemm_k256_image_bytes(int N, int K) {
    if (K != KD || N <= 0 || (N % CW) != 0) return -1;
    return (long)(N / CW) * CHUNK_BYTES;
}

extern "C" int gom_gemm_k256_image(const void* w_planes, long w_plane_stride, int ldw, const float* w_inv_scale,
                                   const float* bias, int N, int K, void* image, long image_bytes, void* stream) {
    GOM_CHECK_ARG(w_planes && w_inv_scale && image && K == KD && N > 0 && (N % CW) == 0 && ldw >= KD);
    GOM_CHECK_ARG(image_bytes >= gom_gemm_k256_image_bytes(N, K));
    const long total = (long)(N / CW) * CHUNK_FRAGS * 512;
    hipLaunchKernelGGL(k256_image_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)w_planes, w_plane_stride, ldw, w_inv_scale, bias, N, (unsigned short*)image);
    return gom_launch_status();
}

static int g_k256_lines = -1;                            // -1: by M (long problems), 0 / 1: forced (tests, tools)

static int g_k256_interleave = 0;                        // frame-interleaved tile order for periodic residuals: measured 3 % SLOWER (374 vs 364 us), off

extern "C" void gom_gemm_k256_set_lines(int mode) { g_k256_lines = mode; }
extern "C" void gom_gemm_k256_set_interleave(int on) { g_k256_interleave = on; }

static int launch_rows(const float* A, const float* A2, int lda, const void* image, const float* R, int ldr, int r_cols, int r_period,
                       int relu, float* C, int ldc, int M, int N, int K, int col_groups, int* flag, void* stream) {
    GOM_CHECK_ARG(A && image && C && M >= 0 && K == KD && N > 0 && (N % CW) == 0);
    GOM_CHECK_ARG(lda >= KD && (lda % 4) == 0 && ldc >= N && (ldc % 4) == 0 && (!R || (ldr >= r_cols && (ldr % 4) == 0)));
    GOM_CHECK_ARG(((uintptr_t)A % 16) == 0 && (!A2 || ((uintptr_t)A2 % 16) == 0) && ((uintptr_t)C % 16) == 0 &&
                  (!R || ((uintptr_t)R % 16) == 0) && ((uintptr_t)image % 16) == 0);
    GOM_CHECK_ARG(!R || (r_cols > 0 && r_cols <= N && (r_cols % CW) == 0));
    // a periodic residual: at least one wave's rows per period (one wrap per wave), the table below 4 GB (buffer descriptor)
    GOM_CHECK_ARG(r_period == 0 || (R && r_period >= 32 && (long)r_period * ldr * 4 < (1L << 32)));
    if (M == 0) return GOM_OK;
    const int chunks = N / CW, tiles = cdiv(M, BM);
    int groups = col_groups;
    if (groups <= 0) {                                       // ~2 workgroups per CU, whole chunks per group
        groups = cdiv(512, tiles);
        if (groups > chunks) groups = chunks;
        if (groups < 1) groups = 1;
    }
    GOM_CHECK_ARG(groups <= chunks && groups <= 65535);
    RowArgs a{};
    a.A = A; a.A2 = A2; a.img = (const unsigned char*)image; a.R = R; a.C = C; a.flag = flag;
    a.lda = lda; a.ldr = ldr; a.ldc = ldc; a.M = M; a.chunks = chunks; a.cpg = cdiv(chunks, groups);
    a.r_chunks = R ? r_cols / CW : 0; a.relu = relu ? 1 : 0; a.r_period = r_period;
    a.tiles = tiles; a.frames = 1; a.tpf = tiles;
    unsigned ids = (unsigned)tiles;
    if (g_k256_interleave && R && r_period > 0 && M % r_period == 0 && M / r_period > 1 && tiles >= 512 && cdiv(chunks, a.cpg) == 1) {
        a.frames = M / r_period;
        a.tpf = cdiv(tiles, a.frames);
        ids = (unsigned)(cdiv(a.tpf, 8) * 8 * a.frames);
    }
    const dim3 grid(ids, (unsigned)cdiv(chunks, a.cpg));
    // (the attribute is per DEVICE: set on every launch -- a process-wide flag would miss a second GPU; it costs ~1 us)
    hipError_t e = hipFuncSetAttribute((const void*)gemm_k256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)gemm_k256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    // whole-line stores for long problems (>= one round of two workgroups per CU), see the kernel's header
    const bool lines = g_k256_lines < 0 ? tiles >= 512 : g_k256_lines != 0;
    if (lines) hipLaunchKernelGGL(gemm_k256_kernel<true>, grid, dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(gemm_k256_kernel<false>, grid, dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    return gom_launch_status();
}

extern "C" int gom_gemm_k256_rp_f32(const float* A, const float* A2, int lda, const void* image, const float* R, int ldr,
                                    int r_cols, int r_period, int relu, float* C, int ldc, int M, int N, int K, int col_groups,
                                    int* flag, void* stream) {
    return launch_rows(A, A2, lda, image, R, ldr, r_cols, r_period, relu, C, ldc, M, N, K, col_groups, flag, stream);
}

extern "C" int gom_gemm_k256_f32(const float* A, const float* A2, int lda, const void* image, const float* R, int ldr,
                                 int r_cols, int relu, float* C, int ldc, int M, int N, int K, int col_groups, int* flag,
                                 void* stream) {
    return gom_gemm_k256_rp_f32(A, A2, lda, image, R, ldr, r_cols, 0, relu, C, ldc, M, N, K, col_groups, flag, stream);
}
