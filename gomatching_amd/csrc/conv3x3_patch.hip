// 3x3 / stride 1 / pad 1 convolution of the ResNet bottlenecks with the INPUT PATCH RESIDENT in LDS (f16x3 split, fp32-class):
//
//   Y[b, y, x, :] = act( sum_taps X[b, y + dy - 1, x + dx - 1, :] . W[:, dy, dx, :]^T * scale + shift )      NHWC fp32
//
// = conv2 of every Detectron2 v0.6 BottleneckBlock behind /root/reference/gomatching/modeling/meta_arch/gom_lstmatcher.py:42-61
// (13 of ResNet-50's 16 3x3 convolutions; the three strided ones stay on the implicit-GEMM kernel of gemm_f16x3.hip).
//
// The implicit-GEMM kernel builds its A tile per k-tile: every input pixel is loaded from L2, split into two fp16 planes and
// stored to LDS NINE times (once per tap), between two barriers, with the matrix pipe of that workgroup idle -- removing just
// that staging (diagnostic build, wrong results) took the kernel from 246 to 182 us per launch.  Here a workgroup owns an 8 x 16
// block of output pixels of one frame; the 10 x 18 input pixels under it are loaded ONCE per 64-channel chunk as whole
// 256-byte pieces, split once, and stay in LDS as two fp16 planes (128-byte pixels, 16-byte slots XOR-swizzled by the pixel:
// conflict-free ds_read_b128 for any 16 consecutive pixels); the A operand of tap (dy, dx) is the same patch read at a shifted
// address.  Only the weights move in the k-loop: a fragment-linear image (gom_conv3x3_patch_image) streamed through a
// two-stage LDS ring by MUBUF LDS-DMA -- no registers, no VALU, ONE barrier per k-tile.  The next chunk's patch is requested
// while the current one is multiplied.  v_mfma_f32_16x16x32_f16: a fragment = 16 pixels of one row x 32 channels of one tap.
// 78 KB of LDS: two workgroups per CU.
// Summation order: 64-channel chunk, tap, 32-channel k-step (the tile kernel: tap, k-step) -- fp32-class, not its bits for C > 64.
#include "common.h"

namespace {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int TH = 8, TW = 16;                           // output pixels of a workgroup: 8 rows x 16 columns = 128 GEMM rows
constexpr int PH = TH + 2, PW = TW + 2, NPIX = PH * PW;  // input patch: 10 x 18 = 180 pixels
constexpr int KC = 64;                                   // channels resident at a time
constexpr int PRS = KC * 2;                              // patch pixel stride per plane: 128 B = 8 sixteen-byte slots, slot s at s ^ ((pixel >> 1) & 7)
constexpr int P_PLANE = NPIX * PRS;                      // 23 040 B
constexpr int FRAG = 1024;                               // one MFMA operand fragment
constexpr int P_UNITS = (NPIX * 16 + 255) / 256;         // float4 units per thread per chunk: 12
constexpr int KT_PER_CHUNK = 9 * (KC / 32);              // 18 k-tiles of 32

struct PArgs {
    const float* X;
    const unsigned char* img;                            // fragment-linear weight image (conv3x3_image_kernel)
    const float* wscale;
    const float* scale;
    const float* shift;
    float* Y;
    int* flag;
    int B, H, W, C, N, relu, tiles_x, tiles_y;
};

__device__ __forceinline__ void split4(const f32x4 v, u32x2& p0, u32x2& p1) {
    unsigned int a0, a1, b0, b1;
    gom_split2_f16(v[0], v[1], a0, a1);
    gom_split2_f16(v[2], v[3], b0, b1);
    p0 = u32x2{a0, b0};
    p1 = u32x2{a1, b1};
}

template <int BN>
__global__ __launch_bounds__(256, 2) void conv3x3_patch_kernel(const PArgs p) {
    constexpr int WN = BN / 2, NT = WN / 16, MT = 4;         // 2 x 2 waves: 4 pixel rows x WN columns each
    constexpr int KT_FRAGS = (BN / 16) * 2;                  // a k-tile of weights: column groups x planes
    constexpr int KT_BYTES = KT_FRAGS * FRAG;
    constexpr int KPS = 128 / BN;                            // k-tiles per ring stage: a stage is always 16 KB = one barrier
    constexpr int ST_FRAGS = KT_FRAGS * KPS, ST_BYTES = ST_FRAGS * FRAG;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ps = smem;                                // [2][180][128 B]
    unsigned char* Ws = smem + 2 * P_PLANE;                  // [2 stages][ST_FRAGS][1 KB]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int fn = lane & 15, fg = lane >> 4;

    const int tiles_n = p.N / BN;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tn = bid % tiles_n;
    int tm = bid / tiles_n;
    const int tx = tm % p.tiles_x;
    tm /= p.tiles_x;
    const int ty = tm % p.tiles_y, b = tm / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BN;

    constexpr unsigned RANGE = 0x80000000u, INVALID = 0xC0000000u;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (int)RANGE, 0x00020000);
    const int chunks = p.C / KC, total = chunks * KT_PER_CHUNK;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.img + (size_t)tn * total * KT_BYTES), 0,
                                                                         total * KT_BYTES, 0x00020000);

    // patch unit u = tid + 256 i: pixel u >> 4 of the patch, channels 4 (u & 15) .. + 3 of the chunk
    unsigned p_off[P_UNITS];
    int p_dst[P_UNITS];
#pragma unroll
    for (int i = 0; i < P_UNITS; ++i) {
        const int u = tid + i * 256, pp = u >> 4, q = u & 15;
        const int py = pp / PW, px = pp - py * PW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool ok = u < NPIX * 16 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        p_off[i] = ok ? (unsigned)((((b * p.H + gy) * p.W + gx) * p.C + q * 4) * 4) : INVALID;   // outside: zeros (the padding)
        p_dst[i] = u < NPIX * 16 ? pp * PRS + (((q >> 1) ^ ((pp >> 1) & 7)) << 4) + (q & 1) * 8 : -1;
    }
    f32x4 p_reg[P_UNITS];
    auto load_patch = [&](int cc) {
#pragma unroll
        for (int i = 0; i < P_UNITS; ++i)
            p_reg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(p_off[i] + (unsigned)cc * (KC * 4)), 0, 0));
    };
    float amax = 0.f;
    auto store_patch = [&]() {
#pragma unroll
        for (int i = 0; i < P_UNITS; ++i) {
            if (p_dst[i] < 0) continue;
            u32x2 h0, h1;
#pragma unroll
            for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(p_reg[i][e]));
            split4(p_reg[i], h0, h1);
            *reinterpret_cast<u32x2*>(Ps + p_dst[i]) = h0;
            *reinterpret_cast<u32x2*>(Ps + P_PLANE + p_dst[i]) = h1;
        }
    };
    auto dma_W = [&](int st) {                               // stage st of the launch (k-tiles KPS st ..) -> ring slot st & 1
        unsigned char* dst = Ws + (st & 1) * ST_BYTES;
#pragma unroll
        for (int f = 0; f < ST_FRAGS / 4; ++f)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dst + (wave + 4 * f) * FRAG), 16,
                                                     (int)((unsigned)st * ST_BYTES + (wave + 4 * f) * FRAG + lane * 16), 0, 0, 0);
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // operand lane (n = lane & 15, kg = lane >> 4): pixel n of a 16-pixel row (A) / column n of a 16-column group (B), k 8 kg .. + 7
    const int pp_base = (wr * 4) * PW + fn;                  // the lane's patch pixel for tile row 4 wr, tap (0, 0)
    // A fragments of two consecutive k-tiles in two register sets: the reads of the next k-tile are issued BEFORE the barrier and
    // the products that separate it from its own (the patch does not change inside a chunk, and 18 k-tiles per chunk is even)
    half8 afx[2][MT], afy[2][MT];
    auto load_A = [&](int r, half8 (&af)[2][MT]) {           // k-tile r of the chunk: tap r >> 1 = 3 dy + dx, channels 32 (r & 1) .. + 31
        const int tap = r >> 1, dy = tap / 3, dx = tap - 3 * dy;
        const int slot = (r & 1) * 4 + fg;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int pp = pp_base + (i + dy) * PW + dx;
            const unsigned char* a = Ps + pp * PRS + ((slot ^ ((pp >> 1) & 7)) << 4);
            af[0][i] = *reinterpret_cast<const half8*>(a);
            af[1][i] = *reinterpret_cast<const half8*>(a + P_PLANE);
        }
    };
    auto mma = [&](int kt, const half8 (&af)[2][MT]) {       // k-tile kt of the launch (ring slot (kt / KPS) & 1) against af
        const unsigned char* w_base = Ws + ((kt / KPS) & 1) * ST_BYTES + (kt % KPS) * KT_BYTES + (wc * NT * 2) * FRAG + lane * 16;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const half8 b0 = *reinterpret_cast<const half8*>(w_base + (2 * j) * FRAG);
            const half8 b1 = *reinterpret_cast<const half8*>(w_base + (2 * j + 1) * FRAG);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                f32x4 c = acc[i][j];                         // smallest terms first
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1][i], b0, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0][i], b1, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0][i], b0, c, 0, 0, 0);
                acc[i][j] = c;
            }
        }
    };
    bool fresh = chunks > 1;                                 // the 12 youngest requests are a patch that may stay in flight
    // This wave's share of a stage has landed (a patch requested BEHIND it keeps flying for one more stage), its LDS writes are
    // done; then everybody's: nobody still reads the other slot / the old patch.  (Not __syncthreads(): its fence waits for
    // vmcnt(0), i.e. for the patch in flight.)
    auto stage_sync = [&](int st) {
        if (fresh) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        fresh = false;
        if ((st + 1) * KPS < total) dma_W(st + 1);
    };

    load_patch(0);
    dma_W(0);
    store_patch();
    if (fresh) load_patch(1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the first patch is in place
    load_A(0, afx);
    for (int kt = 0; kt < total; kt += 2) {                  // pairs of k-tiles never straddle a chunk
        const int cc = kt / KT_PER_CHUNK, r = kt - cc * KT_PER_CHUNK;
        stage_sync(kt / KPS);
        load_A(r + 1, afy);
        mma(kt, afx);
        if (KPS == 1) stage_sync(kt + 1);
        if (r + 2 < KT_PER_CHUNK) load_A(r + 2, afx);
        mma(kt + 1, afy);
        if (r + 2 == KT_PER_CHUNK && kt + 2 < total) {       // the next k-tile opens a chunk: its patch replaces this one
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            store_patch();
            if (cc + 2 < chunks) {
                load_patch(cc + 2);
                fresh = true;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            load_A(0, afx);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- epilogue: 32-pixel slabs (two tile rows) through the now free patch area to whole 16-byte pieces of NHWC rows ----
    int bad = !(amax <= 65504.f);                            // an input beyond fp16's range (gemm_f16x3.hip contract)
    const float relu_lo = p.relu ? 0.f : -INFINITY;
    constexpr int ES = WN + 4;
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * ES);
    constexpr int C4 = WN / 4, RPI = 64 / C4;
    const int c4 = (lane % C4) * 4;
    const int n = n0 + wc * WN + c4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
    if (p.wscale) sc = sc * *reinterpret_cast<const f32x4*>(p.wscale + n);                     // exact: a power of two
    if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) stage[(16 * ii + 4 * fg + r) * ES + j * 16 + fn] = acc[2 * s + ii][j][r];
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): the slab is wave-private
#pragma unroll
        for (int t = 0; t < 32 / RPI; ++t) {
            const int row = t * RPI + lane / C4;             // pixel of the slab: tile row 2 s + (row >> 4) of the wave, column row & 15
            const int gy = y0 + wr * 4 + 2 * s + (row >> 4), gx = x0 + (row & 15);
            f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * ES + c4);
            v = v * sc + sh;
            const bool bad_v = !(fabsf(v[0]) <= 3.4e38f) | !(fabsf(v[1]) <= 3.4e38f) | !(fabsf(v[2]) <= 3.4e38f) |
                               !(fabsf(v[3]) <= 3.4e38f);  // before the activation: fmaxf(NaN, 0) = 0
            v[0] = fmaxf(v[0], relu_lo); v[1] = fmaxf(v[1], relu_lo);
            v[2] = fmaxf(v[2], relu_lo); v[3] = fmaxf(v[3], relu_lo);
            if (gy < p.H && gx < p.W) {
                bad |= bad_v;
                *reinterpret_cast<f32x4*>(p.Y + ((size_t)(b * p.H + gy) * p.W + gx) * p.N + n) = v;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // reads done before the next slab overwrites
    }
    if (bad && p.flag) atomicOr(p.flag, 1);
}

template <int BN>
int launch(const PArgs& a, hipStream_t s) {
    const long wgs = (long)a.B * a.tiles_y * a.tiles_x * (a.N / BN);
    constexpr int lds = 2 * P_PLANE + 2 * 16 * FRAG;         // patch planes + two 16 KB weight stages (the epilogue slabs fit inside)
    static_assert(4 * 32 * (BN / 2 + 4) * 4 <= 2 * P_PLANE, "epilogue slabs");
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_patch_kernel<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    hipLaunchKernelGGL(conv3x3_patch_kernel<BN>, dim3((unsigned)wgs), dim3(256), lds, s, a);
    return gom_launch_status();
}

// Fragment-linear weight image: column tile tn (BN columns), k-tile g = (chunk cc, tap, half kk), column group j, plane p -> 1 KB:
// element e of lane l = plane p of Ws[BN tn + 16 j + (l & 15)][tap C + 64 cc + 32 kk + 8 (l >> 4) + e]   (Ws: gom_split_f16x2 planes)
__global__ __launch_bounds__(256) void conv3x3_image_kernel(const unsigned short* __restrict__ planes, long plane_stride, int ldw,
                                                            int C, int N, int BN, unsigned short* __restrict__ img) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = 2L * N * 9 * C;
    if (idx >= total) return;
    const int e = (int)(idx & 7), l = (int)((idx >> 3) & 63);
    long f = idx >> 9;                                       // fragment index
    const int st_frags = (BN / 16) * 2;
    const int pl = (int)(f & 1), j = (int)((f % st_frags) >> 1);
    f /= st_frags;
    const int kts = (C / KC) * KT_PER_CHUNK;
    const int g = (int)(f % kts), tn = (int)(f / kts);
    const int cc = g / KT_PER_CHUNK, r = g - cc * KT_PER_CHUNK, tap = r >> 1, kk = r & 1;
    img[idx] = planes[pl * plane_stride + (size_t)(BN * tn + 16 * j + (l & 15)) * ldw + tap * C + KC * cc + 32 * kk + 8 * (l >> 4) + e];
}

}  // namespace

extern "C" int gom_conv3x3_patch_supported(int Cin, int Cout) { return Cin >= KC && Cin % KC == 0 && Cout >= 64 && Cout % 64 == 0; }

extern "C" long gom_conv3x3_patch_image_bytes(int Cin, int Cout) {
    return gom_conv3x3_patch_supported(Cin, Cout) ? 2L * Cout * 9 * Cin * 2 : -1;
}

extern "C" int gom_conv3x3_patch_image(const void* w_planes, long w_plane_stride, int ldw, int Cin, int Cout, void* image,
                                       long image_bytes, void* stream) {
    GOM_CHECK_ARG(w_planes && image && gom_conv3x3_patch_supported(Cin, Cout) && ldw >= 9 * Cin);
    GOM_CHECK_ARG(image_bytes >= gom_conv3x3_patch_image_bytes(Cin, Cout));
    const long total = 2L * Cout * 9 * Cin;
    hipLaunchKernelGGL(conv3x3_image_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)w_planes, w_plane_stride, ldw, Cin, Cout, (Cout % 128) == 0 ? 128 : 64,
                       (unsigned short*)image);
    return gom_launch_status();
}

extern "C" int gom_conv3x3_patch_f32_f16x3(const float* X, const void* image, const float* wscale, const float* scale,
                                           const float* shift, int relu, float* Y, int B, int H, int Wd, int Cin, int Cout,
                                           int* flag, void* stream) {
    GOM_CHECK_ARG(X && image && Y && B > 0 && H > 0 && Wd > 0);
    GOM_CHECK_ARG(gom_conv3x3_patch_supported(Cin, Cout));
    GOM_CHECK_ARG((long)B * H * Wd * Cin < (1L << 29) && (long)B * H * Wd * Cout < (1L << 31));
    GOM_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0 && ((uintptr_t)image % 16) == 0);
    PArgs a{};
    a.X = X; a.img = (const unsigned char*)image; a.wscale = wscale; a.scale = scale;
    a.shift = shift; a.Y = Y; a.flag = flag; a.B = B; a.H = H; a.W = Wd; a.C = Cin; a.N = Cout; a.relu = relu ? 1 : 0;
    a.tiles_x = cdiv(Wd, TW); a.tiles_y = cdiv(H, TH);
    return (Cout % 128) == 0 ? launch<128>(a, (hipStream_t)stream) : launch<64>(a, (hipStream_t)stream);
}
