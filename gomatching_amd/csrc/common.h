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
