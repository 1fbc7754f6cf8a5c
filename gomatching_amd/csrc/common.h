// Shared helpers for the gfx950 kernels of the GoMatching hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gomatching_hip.h"

#define GOM_CHECK_ARG(cond)                      \
    do {                                         \
        if (!(cond)) return GOM_ERR_INVALID_ARG; \
    } while (0)

// Launch-error reporting: unlike the reference (which only printf()s cudaGetLastError,
// ms_deform_im2col_cuda.cuh:948-952) every entry point returns the HIP error to the caller.
static inline int gom_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GOM_OK : (GOM_ERR_HIP_BASE + (int)e);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// The f16x3 split of two fp32 values, x = h0 + h1 with h0 = fp16(x), h1 = fp16(x - h0) (gemm_f16x3.hip): `hi` / `lo` = the packed
// (x, y) halves of the two planes.  The residual x - float(h0) is ONE v_fma_mix_f32 per value (the fp16 operand is widened
// inside the instruction, times -1.0, plus x: exact product, one rounding) instead of v_cvt_f32_f16 + v_sub_f32: the same bits
// (tests/test_ops_gpu.py compares the planes with a numpy statement of the split), a third fewer VALU instructions -- which is
// what every kernel that splits activations in its loop pays with (4 cycles per wave-instruction, matrix pipe idle meanwhile).
__device__ __forceinline__ void gom_split2_f16(float x, float y, unsigned int& hi, unsigned int& lo) {
    typedef _Float16 gom_h2 __attribute__((ext_vector_type(2)));
    typedef float gom_f2 __attribute__((ext_vector_type(2)));
    const gom_f2 v = {x, y};
    hi = __builtin_bit_cast(unsigned int, __builtin_convertvector(v, gom_h2));
    float rx, ry;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rx) : "v"(hi), "v"(x));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ry) : "v"(hi), "v"(y));
    const gom_f2 r = {rx, ry};
    lo = __builtin_bit_cast(unsigned int, __builtin_convertvector(r, gom_h2));
}

// A launch's weight image towards this XCD's L2, once, at the start.  Between two launches that use it the image (0.3 - 3.4 MB, last
// read a layer or a step ago) is in HBM; every workgroup of a round then streams the same stage at the same time, and a ring's one or
// two stages of lookahead do not cover a miss per stage (dec_attn2.hip: inter + raw 76 us warm, 100 us behind a 768 MB fill, 88 us with
// this).  Workgroups go to the XCDs round-robin (1-D grids): the first-round workgroups of an XCD (at most 32) touch one line in 128
// bytes of the image each, a slice per workgroup.  The loads' results are never used; they are the oldest vector-memory operations of
// the wave, so the first counted wait of the kernel covers them.  Decoder launches only (one round of one-per-CU workgroups, 200-250 of
// them): in the encoder's and the backbone's row-resident kernels -- nine or more rounds, the image cold for the first only -- the
// same call measured -0.5 % frames/s (same box, three alternating pairs) and is not made.
// K loads per lane, unconditional (addresses clamped), unrolled and ORDINARY: the compiler then counts them like any other load.  (A
// loop of unknown length or a branch makes it wait for every outstanding load at the next use of any loaded value; a `volatile` load is
// followed by a wait for ALL of them on the spot -- either way the prologue's rows wait for the image's lines, +7k cycles.)  The
// values land in pf[]; gom_prefetch_done(pf) -- a use that generates no code -- goes where the kernel has to wait for its first
// loads anyway.  K x nthreads lines per workgroup are covered; a launch of few workgroups covers less.
template <int K>
__device__ __forceinline__ void gom_prefetch_image(const void* img, unsigned bytes, unsigned tid, unsigned nthreads, unsigned (&pf)[K]) {
    const unsigned first = gridDim.x < 256u ? gridDim.x : 256u;
    const unsigned nsl = (first + 7) / 8, sl = (blockIdx.x < first ? blockIdx.x : 0u) / 8;
    const unsigned lines = bytes / 128;
    const unsigned per = (lines + nsl - 1) / nsl;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        unsigned l = tid + k * nthreads;
        l = l < per ? l : per - 1;
        unsigned line = sl * per + l;
        line = line < lines ? line : lines - 1;
        pf[k] = *reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(img) + (size_t)line * 128);
    }
}
template <int K>
__device__ __forceinline__ void gom_prefetch_done(const unsigned (&pf)[K]) {
#pragma unroll
    for (int k = 0; k < K; ++k) asm volatile("" ::"v"(pf[k]));
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// eight fp32 values -> one MFMA operand fragment piece per plane (hi, lo)
__device__ __forceinline__ void gom_split8_f16(const f32x4 a, const f32x4 b, half8& p0, half8& p1) {
    typedef unsigned int gom_u4 __attribute__((ext_vector_type(4)));
    unsigned int l0, l1, l2, l3, h0, h1, h2, h3;
    gom_split2_f16(a[0], a[1], l0, h0);
    gom_split2_f16(a[2], a[3], l1, h1);
    gom_split2_f16(b[0], b[1], l2, h2);
    gom_split2_f16(b[2], b[3], l3, h3);
    p0 = __builtin_bit_cast(half8, (gom_u4{l0, l1, l2, l3}));
    p1 = __builtin_bit_cast(half8, (gom_u4{h0, h1, h2, h3}));
}

// A wave's 32 rows x 256 fp32 -> the two fp16 planes of its MFMA operand fragments: lane (r = lane & 31, h = lane >> 5) ends up
// with floats 16 s + 8 h .. + 7 of row r for every k-step s, xf[plane][s].  Loading that layout straight from memory makes every
// wave-instruction touch 32 lines (16-byte pieces of 32 rows), and the texture-address unit pays per LINE (~3.6 cycles,
// profiles/r03_msda_ta_counters.txt): the 128 such instructions of a 128-row tile cost it ~14 000 cycles -- which is what the
// prologue of every row-resident kernel measured (ffn_fused.hip: 14.2k).  Here the rows are loaded as WHOLE lines (a wave-
// instruction = W floats of 256 / W rows: 8 lines) and change layout in a wave-private LDS scratch of 32 x W fp32 (16-byte
// pieces XOR-swizzled by the row: conflict-free both ways), W of a row's 256 floats at a time.  `row_a(r)` / `row_b(r)` return
// the address of row r (0..31) of the wave; ADD: the fragments are those of row_a + row_b.  `amax` = running largest magnitude
// (the f16x3 range check).  The wave barriers keep the compiler from moving the exchange into divergent code.
// `after_first_loads()` runs once, behind the first part's loads (where a kernel requests its first weight stage: the rows are
// then ahead of it in the memory queue and the first split runs while the weights are still on their way).
// K32 = true: the fragments of the 16x16x32 MFMA shape instead -- lane (n = lane & 15, kg = lane >> 4) ends up with floats
// 32 s + 8 kg .. + 7 of row 16 R + n for every 32-wide k-step s and row group R, xf[plane][8 R + s].
template <int W, bool ADD, bool K32, typename FA, typename FB, typename FH>
__device__ __forceinline__ void gom_rows_to_fragments_t(FA row_a, FB row_b, float* scratch, int lane, half8 (&xf)[2][16], float& amax,
                                                        FH after_first_loads) {
    constexpr int PC = W / 4;                                // 16-byte pieces of a row part
    constexpr int RPI = 64 / PC;                             // rows per wave-instruction
    constexpr int NI = 32 / RPI;                             // wave-instructions per part
    const int pc = lane % PC, r0 = lane / PC;
    const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int part = 0; part < 256 / W; ++part) {
        f32x4 v[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = r0 + RPI * i;
            v[i] = *reinterpret_cast<const f32x4*>(row_a(r) + part * W + pc * 4);
            if constexpr (ADD) v[i] += *reinterpret_cast<const f32x4*>(row_b(r) + part * W + pc * 4);
        }
        if (part == 0) {
            __builtin_amdgcn_sched_barrier(0);
            after_first_loads();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = r0 + RPI * i;
            *reinterpret_cast<f32x4*>(scratch + r * W + ((pc ^ (r & (PC - 1))) << 2)) = v[i];
        }
        __builtin_amdgcn_wave_barrier();
        if constexpr (K32) {
            const int n = lane & 15, kg = lane >> 4;
#pragma unroll
            for (int s = 0; s < W / 32; ++s)
#pragma unroll
                for (int R = 0; R < 2; ++R) {
                    const int row = 16 * R + n, p0 = 8 * s + 2 * kg;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(scratch + row * W + ((p0 ^ (row & (PC - 1))) << 2));
                    const f32x4 b = *reinterpret_cast<const f32x4*>(scratch + row * W + (((p0 + 1) ^ (row & (PC - 1))) << 2));
#pragma unroll
                    for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(a[e]), fabsf(b[e])));
                    gom_split8_f16(a, b, xf[0][8 * R + part * (W / 32) + s], xf[1][8 * R + part * (W / 32) + s]);
                }
        } else {
#pragma unroll
            for (int s = 0; s < W / 16; ++s) {
                const int p0 = 4 * s + 2 * fh;
                const f32x4 a = *reinterpret_cast<const f32x4*>(scratch + fr * W + ((p0 ^ (fr & (PC - 1))) << 2));
                const f32x4 b = *reinterpret_cast<const f32x4*>(scratch + fr * W + (((p0 + 1) ^ (fr & (PC - 1))) << 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(a[e]), fabsf(b[e])));
                gom_split8_f16(a, b, xf[0][part * (W / 16) + s], xf[1][part * (W / 16) + s]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    asm volatile("" : "+v"(amax));                           // decided here (dec_attn.hip: deferred compares spill)
}

template <int W, bool ADD, typename FA, typename FB, typename FH>
__device__ __forceinline__ void gom_rows_to_fragments(FA row_a, FB row_b, float* scratch, int lane, half8 (&xf)[2][16], float& amax,
                                                      FH after_first_loads) {
    gom_rows_to_fragments_t<W, ADD, false>(row_a, row_b, scratch, lane, xf, amax, after_first_loads);
}
