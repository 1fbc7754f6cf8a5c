"""Diagnostic build of the fused FFN kernel: where do a workgroup's cycles go, and what clock does the chip hold under it?

Generates a stamped copy of gomatching_amd/csrc/ffn_fused.hip (s_memtime + s_memrealtime at kernel start, chunk-loop start,
chunk-loop end, kernel end; wave 0 of each workgroup stores them), compiles it beside this file and runs it at the encoder's
shape after two seconds of back-to-back launches.  The product kernel carries no stamps.
    python tools/exp/ffn_clock.py --build      (here, no GPU)       python tools/exp/ffn_clock.py   (GPU box)"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "libffn_clock.so")


def build():
    src = open(os.path.join(ROOT, "gomatching_amd", "csrc", "ffn_fused.hip")).read()
    def once(s, a, b):
        assert s.count(a) == 1, a
        return s.replace(a, b)
    src = once(src, 'template <bool PLAIN, int RG = 2>\n__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(const FfnArgs p) {',
               '__device__ unsigned long long g_stamp[4096 * 8];\n'
               '#define STAMP(i) if (blockIdx.x < 4096 && threadIdx.x == 0) { g_stamp[blockIdx.x * 8 + 2 * (i)] = __builtin_amdgcn_s_memtime(); '
               'g_stamp[blockIdx.x * 8 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); }\n'
               'template <bool PLAIN, int RG = 2>\n__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(const FfnArgs p) {\n    STAMP(0)')
    src = once(src, '    constexpr unsigned OOB = 0x7FFF0000u;', '    STAMP(1)\n    constexpr unsigned OOB = 0x7FFF0000u;')
    src = once(src, '    // ---- epilogue: Y^T (row of X on the lane', '    STAMP(2)\n    // ---- epilogue: Y^T (row of X on the lane')
    src = once(src, '    if (bad && p.flag) atomicOr(p.flag, 1);', '    STAMP(3)\n    if (bad && p.flag) atomicOr(p.flag, 1);')
    # finer: cycles inside the chunk loop spent in the activation conversion and in the end-of-chunk wait + barrier
    src = once(src, '    for (int c = 0; c < p.chunks; ++c) {\n        const int st = c & 1;',
               '    unsigned long long t_conv = 0, t_wait = 0;\n    for (int c = 0; c < p.chunks; ++c) {\n        const int st = c & 1;')
    src = once(src, '        const float* aux = reinterpret_cast<const float*>(smem + st * STAGE_BYTES',
               '        const unsigned long long ta_ = FINE ? __builtin_amdgcn_s_memtime() : 0;\n'
               '        const float* aux = reinterpret_cast<const float*>(smem + st * STAGE_BYTES')
    src = once(src, '        // ---- Y^T += W2[:, chunk] . H^T',
               '        if (FINE) t_conv += __builtin_amdgcn_s_memtime() - ta_;\n        // ---- Y^T += W2[:, chunk] . H^T')
    src = once(src, '        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave\'s share of the next stage has landed\n        __syncthreads();',
               '        const unsigned long long tc_ = FINE ? __builtin_amdgcn_s_memtime() : 0;\n'
               '        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n'
               '        const unsigned long long tv_ = FINE ? __builtin_amdgcn_s_memtime() : 0;\n        __syncthreads();\n'
               '        if (FINE) { t_wait += tv_ - tc_; t_conv += 0; t_bar += __builtin_amdgcn_s_memtime() - tv_; }')
    src = once(src, '    unsigned long long t_conv = 0, t_wait = 0;', '    unsigned long long t_conv = 0, t_wait = 0, t_bar = 0;')
    src = once(src, '    STAMP(2)\n', '    STAMP(2)\n    if (FINE && blockIdx.x < 4096 && (threadIdx.x & 63) == 0) { g_fine[(blockIdx.x * 4 + wave) * 3] = t_conv; '
               'g_fine[(blockIdx.x * 4 + wave) * 3 + 1] = t_wait; g_fine[(blockIdx.x * 4 + wave) * 3 + 2] = t_bar; }\n')
    src = once(src, '__device__ unsigned long long g_stamp[4096 * 8];', '__device__ unsigned long long g_stamp[4096 * 8];\n__device__ unsigned long long g_fine[4096 * 12];\n'
               '#ifndef FINE\n#define FINE 0\n#endif')
    src += ('\nextern "C" int ffn_fine_read(unsigned long long* host) {\n'
            '    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fine), sizeof(unsigned long long) * 4096 * 12);\n}\n')
    src += ('\nextern "C" int ffn_clock_read(unsigned long long* host) {\n'
            '    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 4096 * 8);\n}\n')
    src = once(src, '#define FFN_DMA(i) dma_fragment(rs_img, nsrc + (i) * 4 * FRAG, ndst + (i) * 4 * FRAG);',
               '#if NODMA\n#define FFN_DMA(i)\n#else\n#define FFN_DMA(i) dma_fragment(rs_img, nsrc + (i) * 4 * FRAG, ndst + (i) * 4 * FRAG);\n#endif')
    src = once(src, '#ifndef FINE\n#define FINE 0\n#endif', '#ifndef FINE\n#define FINE 0\n#endif\n#ifndef NODMA\n#define NODMA 0\n#endif')
    gen = os.path.join(HERE, "_ffn_clock_gen.hip")
    open(gen, "w").write(src)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-I", os.path.join(ROOT, "gomatching_amd", "csrc"),
                           "-I", os.path.join(ROOT, "include"), gen, "-o", SO])
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DFINE=1", "-I", os.path.join(ROOT, "gomatching_amd", "csrc"),
                           "-I", os.path.join(ROOT, "include"), gen, "-o", SO.replace(".so", "_fine.so")])
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DNODMA=1", "-I", os.path.join(ROOT, "gomatching_amd", "csrc"),
                           "-I", os.path.join(ROOT, "include"), gen, "-o", SO.replace(".so", "_nodma.so")])
    os.remove(gen)


def main(fine=False, nodma=False):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from gomatching_amd import ops
    so = ctypes.CDLL(SO.replace(".so", "_nodma.so") if nodma else SO.replace(".so", "_fine.so") if fine else SO)
    if nodma:
        print("NODMA build: the loop re-reads chunk 0's weights (wrong results on purpose): the loop without its weight stream")
    so.gom_ffn_fused_ln_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    so.ffn_clock_read.argtypes = [ctypes.c_void_p]
    dev, F = "cuda", 1024
    g = torch.Generator().manual_seed(0)
    w1 = (torch.randn((F, 256), generator=g) * 0.05).to(dev); b1 = torch.randn((F,), generator=g).to(dev) * 0.1
    w2 = (torch.randn((256, F), generator=g) * 0.05).to(dev); b2 = torch.randn((256,), generator=g).to(dev) * 0.1
    ga = torch.ones((256,), device=dev); be = torch.zeros((256,), device=dev)
    ffn = ops.FusedFFN(w1, b1, w2, b2, ga, be)
    for M in (297368, 20000):
        x = torch.randn((M, 256), generator=g).to(dev)
        y = torch.empty_like(x)
        ref = ops.ffn_fused_ln(x, ffn)
        flag = torch.zeros((1,), dtype=torch.int32, device=dev)

        def run():
            rc = so.gom_ffn_fused_ln_f32(x.data_ptr(), 256, ffn.image.data_ptr(), ffn.inv2.data_ptr(), ffn.b2.data_ptr(), ffn.gamma.data_ptr(),
                                         ffn.beta.data_ptr(), ffn.eps, y.data_ptr(), 256, M, 256, F, flag.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        run(); torch.cuda.synchronize()
        assert nodma or torch.equal(y, ref), "stamped build differs from the product kernel"
        import time
        t0 = time.time()
        while time.time() - t0 < 2.0:
            for _ in range(50):
                run()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        buf = np.zeros((4096, 4, 2), dtype=np.uint64)
        assert so.ffn_clock_read(buf.ctypes.data) == 0
        n = min(4096, (M + 127) // 128)
        s = buf[:n].astype(np.int64)
        cyc = s[:, :, 0]; real = s[:, :, 1]
        loop_c = cyc[:, 2] - cyc[:, 1]; loop_r = real[:, 2] - real[:, 1]
        clk = np.median(loop_c / np.maximum(loop_r, 1)) * 100.0          # MHz: s_memrealtime ticks at 100 MHz
        chunks = F // 32
        print("M %d: launch %.1f us (stamped build, 10 back-to-back); in-kernel clock %.0f MHz (median over %d workgroups)" % (M, us, clk, n))
        print("   per workgroup, median cycles: prologue %d | chunk loop %d = %d per chunk (MFMA floor 96 x 32 = 3072: %.0f %% matrix-pipe busy) | epilogue %d | total %d" % (
            np.median(cyc[:, 1] - cyc[:, 0]), np.median(loop_c), np.median(loop_c) / chunks, 100.0 * 3072 * chunks / np.median(loop_c),
            np.median(cyc[:, 3] - cyc[:, 2]), np.median(cyc[:, 3] - cyc[:, 0])))
        if fine:
            fb = np.zeros((4096, 4, 3), dtype=np.uint64)
            so.ffn_fine_read.argtypes = [ctypes.c_void_p]
            assert so.ffn_fine_read(fb.ctypes.data) == 0
            f = fb[:n].astype(np.int64)
            print("   FINE build (stamps inside the loop perturb it): per chunk, median over waves: conversion %d | wait for the next stage's DMA %d | barrier %d cycles" % (
                np.median(f[:, :, 0]) / chunks, np.median(f[:, :, 1]) / chunks, np.median(f[:, :, 2]) / chunks))
        tot = np.median(cyc[:, 3] - cyc[:, 0])
        print("   rounds of workgroups %.2f x %d cycles / clock = %.1f us of the launch" % (n / 256.0, tot, n / 256.0 * tot / clk))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main(False)
        main(True)
        main(False, nodma=True)
