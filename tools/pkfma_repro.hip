// Stand-alone isolation of the tracker determinism issue ("Tracker determinism"): a victim kernel whose only
// arithmetic is v_pk_fma_f32 (the packed-fp32 FMA hipcc's SLP vectoriser emits for adjacent fmaf chains, with the op_sel
// modifiers it uses to broadcast one operand) runs on a high-priority stream while an "aggressor" kernel made of ONE kind of
// instruction saturates the chip from another stream; every victim result is compared with its idle-GPU result.
//   hipcc -O3 --offload-arch=gfx950 tools/pkfma_repro.hip -o gpurun_out/pkfma_repro && gpurun_out/pkfma_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// victim: acc(lo,hi) += x(lo,hi) * y.lo   -- MODE 0: packed with op_sel_hi:[1,0,1] (what the tracker kernel had),
// MODE 1: packed, default modifiers (y.lo for lo, y.hi for hi), MODE 2: two scalar v_fmac_f32
template <int MODE>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    f32x2 x = {in[2 * t], in[2 * t + 1]};
    f32x2 y = {in[2 * t + 1] * 0.5f, in[2 * t] * 0.25f};
    f32x2 acc = {0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "v"(y));
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
        if (MODE == 2) {
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc.x) : "v"(x.x), "v"(y.x));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc.y) : "v"(x.y), "v"(y.x));
        }
        asm volatile("v_mul_f32 %0, 0x3f7fbe77, %0" : "+v"(x.x));      // keep the operands moving (x *= 0.999)
        asm volatile("v_mul_f32 %0, 0x3f7fbe77, %0" : "+v"(x.y));
    }
    out[2 * t] = acc.x;
    out[2 * t + 1] = acc.y;
}

// aggressors: long loops of one instruction kind on every SIMD of the chip
template <int KIND>
__global__ __launch_bounds__(256) void aggressor(float* __restrict__ sink, int iters) {
    f32x2 a = {1.0f + threadIdx.x * 1e-3f, 2.0f}, b = {0.999f, 1.001f}, c = {0.f, 0.f};
    f32x16 m = {};
    s16x8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {8, 7, 6, 5, 4, 3, 2, 1};
    unsigned pk = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(c) : "v"(a));
            if (KIND == 2) asm volatile("v_mov_b64 %0, %1" : "=v"(c) : "v"(a));
            if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(a.x), "v"(a.y));
            if (KIND == 4) m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, m, 0, 0, 0);
            if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(c) : "v"(a), "v"(b));
            if (KIND == 6) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(c.x) : "v"(a.x), "v"(b.x));
            if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(c) : "v"(a), "v"(b));
        }
    }
    if (c.x + c.y + m[0] + (float)pk == 123.456f) sink[0] = c.x;       // keep everything live
}

template <int MODE>
int run_victim(hipStream_t s, const float* in, float* out, int n, int iters) {
    hipLaunchKernelGGL(victim<MODE>, dim3(n / 256), dim3(256), 0, s, in, out, iters);
    return 0;
}

int main(int argc, char** argv) {
    const int n = 256 * 64, iters = 2000, rounds = argc > 1 ? atoi(argv[1]) : 400;
    std::vector<float> h(2 * n);
    for (int i = 0; i < 2 * n; ++i) h[i] = 0.5f + (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f;
    float *in, *out, *sink;
    CK(hipMalloc(&in, 2 * n * 4)); CK(hipMalloc(&out, 2 * n * 4)); CK(hipMalloc(&sink, 64));
    CK(hipMemcpy(in, h.data(), 2 * n * 4, hipMemcpyHostToDevice));
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t sv, sa;
    CK(hipStreamCreateWithPriority(&sv, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    const char* vn[3] = {"v_pk_fma_f32 op_sel_hi:[1,0,1]", "v_pk_fma_f32 (default)", "2 x v_fmac_f32"};
    const char* an[9] = {"v_pk_fma_f32", "v_pk_add_f32", "v_mov_b64", "v_cvt_pk_bf16_f32", "v_mfma_f32_32x32x16_bf16", "v_pk_mul_f32",
                         "v_fmac_f32", "v_pk_fma_f32 op_sel_hi", "(none)"};
    std::vector<float> ref(2 * n), got(2 * n);
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 0) run_victim<0>(sv, in, out, n, iters);
        if (mode == 1) run_victim<1>(sv, in, out, n, iters);
        if (mode == 2) run_victim<2>(sv, in, out, n, iters);
        CK(hipStreamSynchronize(sv));
        CK(hipMemcpy(ref.data(), out, 2 * n * 4, hipMemcpyDeviceToHost));
        for (int kind = 0; kind < 9; ++kind) {
            long bad_runs = 0, bad_lo = 0, bad_hi = 0;
            for (int r = 0; r < rounds; ++r) {
                if (r % 8 == 0 && kind < 8) {
                    const int ai = 20000;
                    if (kind == 0) hipLaunchKernelGGL(aggressor<0>, dim3(2048), dim3(256), 0, sa, sink, ai);
                    if (kind == 1) hipLaunchKernelGGL(aggressor<1>, dim3(2048), dim3(256), 0, sa, sink, ai);
                    if (kind == 2) hipLaunchKernelGGL(aggressor<2>, dim3(2048), dim3(256), 0, sa, sink, ai);
                    if (kind == 3) hipLaunchKernelGGL(aggressor<3>, dim3(2048), dim3(256), 0, sa, sink, ai);
                    if (kind == 4) hipLaunchKernelGGL(aggressor<4>, dim3(2048), dim3(256), 0, sa, sink, ai / 8);
                    if (kind == 5) hipLaunchKernelGGL(aggressor<5>, dim3(2048), dim3(256), 0, sa, sink, ai);
                    if (kind == 6) hipLaunchKernelGGL(aggressor<6>, dim3(2048), dim3(256), 0, sa, sink, ai);
                    if (kind == 7) hipLaunchKernelGGL(aggressor<7>, dim3(2048), dim3(256), 0, sa, sink, ai);
                }
                if (mode == 0) run_victim<0>(sv, in, out, n, iters);
                if (mode == 1) run_victim<1>(sv, in, out, n, iters);
                if (mode == 2) run_victim<2>(sv, in, out, n, iters);
                CK(hipStreamSynchronize(sv));
                CK(hipMemcpy(got.data(), out, 2 * n * 4, hipMemcpyDeviceToHost));
                long b = 0;
                for (int i = 0; i < 2 * n; ++i)
                    if (memcmp(&got[i], &ref[i], 4) != 0) { ++b; if (i & 1) ++bad_hi; else ++bad_lo; }
                bad_runs += b != 0;
            }
            CK(hipDeviceSynchronize());
            printf("victim %-32s beside %-26s: %4ld of %d launches wrong (elements: %ld low halves, %ld high halves)\n", vn[mode],
                   an[kind], bad_runs, rounds, bad_lo, bad_hi);
            fflush(stdout);
        }
    }
    return 0;
}
