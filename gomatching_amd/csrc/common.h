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
