// Experiment: how fast can 1 workgroup per CU stream a 2 MB L2-resident image into LDS by LDS-DMA, as a function of the bytes
// it keeps in flight?  MODE 0: two 65 KB stages, one in flight (the FFN kernel's ring); MODE 1: four 32.5 KB stages, up to
// three in flight; MODE 2: eight 16 KB stages, seven in flight.
#include <hip/hip_runtime.h>
#include <stdint.h>
constexpr int FRAG = 1024;
__device__ __forceinline__ void dma_fragment(__amdgpu_buffer_rsrc_t rs, unsigned byte_offset, unsigned char* lds_frag) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_frag, 16, (int)byte_offset, 0, 0, 0);
}
template <int STAGE_FRAGS, int NST, int WAITN>
__global__ __launch_bounds__(256, 1) void ring_kernel(const unsigned char* img, int total_frags, float* out, int reps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, total_frags * FRAG, 0x00020000);
    const int stages = total_frags / STAGE_FRAGS;
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        auto issue = [&](int k) {
            const int kk = k % stages;
            unsigned char* dst = smem + (k % NST) * STAGE_FRAGS * FRAG;
            for (int f = wave; f < STAGE_FRAGS; f += 4) dma_fragment(rs, (unsigned)(kk * STAGE_FRAGS + f) * FRAG + lane * 16, dst + f * FRAG);
        };
        for (int k = 0; k < NST - 1; ++k) issue(k);
        for (int k = 0; k < stages; ++k) {
            // wait for stage k (issued NST-1 stages ago): younger = (NST-2) stages of pieces
            if (WAITN == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (WAITN == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (WAITN == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            acc += *reinterpret_cast<const float*>(smem + (k % NST) * STAGE_FRAGS * FRAG + lane * 16);   // touch
            __builtin_amdgcn_s_barrier();
            issue(k + NST - 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (acc == 1.2345f) out[0] = acc;
}
extern "C" int run_ring(int mode, const void* img, int total_frags, float* out, int reps, int blocks, void* stream) {
    if (mode == 0) {
        hipFuncSetAttribute((const void*)ring_kernel<64, 2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * FRAG);
        hipLaunchKernelGGL((ring_kernel<64, 2, 0>), dim3(blocks), dim3(256), 2 * 64 * FRAG, (hipStream_t)stream, (const unsigned char*)img, total_frags, out, reps);
    } else if (mode == 1) {
        hipFuncSetAttribute((const void*)ring_kernel<32, 4, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32 * FRAG);
        hipLaunchKernelGGL((ring_kernel<32, 4, 16>), dim3(blocks), dim3(256), 4 * 32 * FRAG, (hipStream_t)stream, (const unsigned char*)img, total_frags, out, reps);
    } else {
        hipFuncSetAttribute((const void*)ring_kernel<16, 8, 24>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16 * FRAG);
        hipLaunchKernelGGL((ring_kernel<16, 8, 24>), dim3(blocks), dim3(256), 8 * 16 * FRAG, (hipStream_t)stream, (const unsigned char*)img, total_frags, out, reps);
    }
    return (int)hipGetLastError();
}
