"""Diagnostic build of the 128x128 tile kernel (gemm_f16x3.hip, three workgroups per CU): where does a wave's lifetime go?
Stamps (s_memtime) around the k-loop's phases -- issue loads + LDS reads + MFMAs | barrier | wait for the loads, split, LDS
stores | barrier -- summed over the k-tiles, plus prologue and epilogue; in-kernel clock from s_memrealtime.  Generated from the
product source; the product kernel carries no stamps.  Shapes: a 3x3 convolution of ResNet-50's layer2 on the BASELINE frames and
the encoder's N = 256 product.
    python tools/exp/tile_clock.py --build   (here)        python tools/exp/tile_clock.py   (GPU box)"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "libtile_clock.so")


def build():
    src = open(os.path.join(ROOT, "gomatching_amd", "csrc", "gemm_f16x3.hip")).read()

    def once(s, a, b):
        assert s.count(a) == 1, a
        return s.replace(a, b)
    src = once(src, 'template <int BM, int BN, int KH, int KW, int OCC>\n__global__ __launch_bounds__(256, OCC) void gemm_f16x3_kernel(const Args p) {',
               '__device__ unsigned long long g_stamp[16384 * 10];\n#define NOW() __builtin_amdgcn_s_memtime()\n'
               'template <int BM, int BN, int KH, int KW, int OCC>\n__global__ __launch_bounds__(256, OCC) void gemm_f16x3_kernel(const Args p) {\n'
               '    const unsigned long long t_start = NOW(), r_start = __builtin_amdgcn_s_memrealtime();\n'
               '    unsigned long long t_comp = 0, t_b1 = 0, t_store = 0, t_b2 = 0, t_loop0 = 0;\n')
    src = once(src, '''        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) {
                load_W(kt + 1);
                load_A(kt + 1, a_even);
            }
            compute();
            __syncthreads();
            if (kt + 1 < nk) {
                store_tile(a_even);
                __syncthreads();
            }
        }''', '''        __syncthreads();
        t_loop0 = NOW();
        for (int kt = 0; kt < nk; ++kt) {
            const unsigned long long t0_ = NOW();
            if (kt + 1 < nk) {
                load_W(kt + 1);
                load_A(kt + 1, a_even);
            }
            compute();
            const unsigned long long t1_ = NOW();
            __syncthreads();
            const unsigned long long t2_ = NOW();
            t_comp += t1_ - t0_; t_b1 += t2_ - t1_;
            if (kt + 1 < nk) {
                store_tile(a_even);
                const unsigned long long t3_ = NOW();
                __syncthreads();
                t_store += t3_ - t2_; t_b2 += NOW() - t3_;
            }
        }''')
    src = once(src, '    // ---- epilogue: y = acc*scale + shift (+ residual) (ReLU) --------------------------------------',
               '    const unsigned long long t_loop1 = NOW();\n    // ---- epilogue')
    src = once(src, '        if (bad && p.flag) atomicOr(p.flag, 1);              // Inf / NaN: an operand left fp16\'s range (or came in bad)\n        return;',
               '        if (blockIdx.x < 16384 && tid == 0) {\n            unsigned long long* o = g_stamp + blockIdx.x * 10;\n'
               '            o[0] = t_loop0 - t_start; o[1] = t_loop1 - t_loop0; o[2] = NOW() - t_loop1; o[3] = t_comp; o[4] = t_b1; o[5] = t_store; o[6] = t_b2;\n'
               '            o[7] = NOW() - t_start; o[8] = __builtin_amdgcn_s_memrealtime() - r_start;\n        }\n'
               '        if (bad && p.flag) atomicOr(p.flag, 1);\n        return;')
    src += ('\nextern "C" int tile_clock_read(unsigned long long* host) {\n'
            '    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 16384 * 10);\n}\n')
    gen = os.path.join(HERE, "_tile_clock_gen.hip")
    open(gen, "w").write(src)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
                           "-I", os.path.join(ROOT, "gomatching_amd", "csrc"), "-I", os.path.join(ROOT, "include"), gen, "-o", SO])
    os.remove(gen)


def main():
    import time
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from gomatching_amd import ops
    ops.GEMM_MODE = "f16x3"
    so = ctypes.CDLL(SO)
    vp, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
    so.gom_gemm_f32_f16x3.argtypes = [vp, vp, I, vp, L, I, vp, vp, vp, vp, I, I, I, vp, I, I, I, I, vp, vp]
    so.gom_conv2d_nhwc_f32_f16x3.argtypes = [vp, vp, L, I, vp, vp, vp, vp, I, vp, I, I, I, I, I, I, I, I, I, vp, L, I, vp, vp]
    so.tile_clock_read.argtypes = [vp]
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    flag = torch.zeros((1,), dtype=torch.int32, device=dev)
    st = lambda: torch.cuda.current_stream().cuda_stream
    cases = []
    # 3x3 convolution, ResNet-50 layer2 (128 -> 128 channels) on 8 frames of 1000 x 1778 at stride 8
    x = torch.randn((8, 125, 223, 128), generator=g).to(dev)
    w = ops.prep_conv_weight((torch.randn((128, 3, 3, 128), generator=g) * 0.03).to(dev))
    sc = torch.rand((128,), generator=g).to(dev) + 0.5; sh = torch.randn((128,), generator=g).to(dev)
    y = torch.empty((8, 125, 223, 128), device=dev)
    ref = ops.conv2d_nhwc(x, w, scale=sc, shift=sh, relu=True, stride=1, pad=1)
    pl = w.planes
    cases.append(("conv 3x3 128->128, M = 223 000, K = 1152", 8 * 125 * 223, 128, 1152, y, ref, lambda: so.gom_conv2d_nhwc_f32_f16x3(
        x.data_ptr(), pl.data_ptr(), pl.stride(0), pl.stride(1), w.inv_scale.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, 1, y.data_ptr(),
        8, 125, 223, 128, 128, 3, 3, 1, 1, None, 0, 1, flag.data_ptr(), st())))
    A = torch.randn((297368, 256), generator=g).to(dev)
    W2 = ops.split_weight((torch.randn((256, 256), generator=g) / 16).to(dev), kind="f16x3")
    b2 = torch.randn((256,), generator=g).to(dev)
    y2 = torch.empty((297368, 256), device=dev)
    ref2 = ops.gemm(A, W2, bias=b2)
    p2 = W2.planes
    cases.append(("GEMM M = 297 368, N = 256, K = 256", 297368, 256, 256, y2, ref2, lambda: so.gom_gemm_f32_f16x3(
        A.data_ptr(), None, 256, p2.data_ptr(), p2.stride(0), p2.stride(1), W2.inv_scale.data_ptr(), None, b2.data_ptr(), None, 0, 0, 0, y2.data_ptr(), 256,
        297368, 256, 256, flag.data_ptr(), st())))
    for name, M, N, K, out, ref_, run in cases:
        assert run() == 0
        torch.cuda.synchronize()
        assert torch.equal(out, ref_), "stamped build differs from the product kernel"
        t0 = time.time()
        while time.time() - t0 < 1.5:
            for _ in range(30):
                run()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        buf = np.zeros((16384, 10), dtype=np.uint64)
        assert so.tile_clock_read(buf.ctypes.data) == 0
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        n = min(16384, tiles)
        s = buf[:n].astype(np.int64)
        nk = (K + 31) // 32
        med = lambda i: float(np.median(s[:, i]))
        clk = float(np.median(s[:, 7] / np.maximum(s[:, 8], 1))) * 100.0
        print("%s: launch %.1f us (stamped build); %d tiles = %.2f rounds of 768; in-kernel clock %.0f MHz" % (name, us, tiles, tiles / 768.0, clk))
        print("   wave 0 of a workgroup, median cycles: prologue %d | k-loop %d | epilogue %d | lifetime %d (%.1f us); MFMA issue of this wave %d (24 x 32 x %d k-tiles), x 3 waves per SIMD = %.0f %% of the lifetime" % (
            med(0), med(1), med(2), med(7), med(7) / clk, 768 * nk, nk, 100.0 * 3 * 768 * nk / med(7)))
        print("   per k-tile: loads issue + LDS reads + MFMAs %d | barrier %d | wait for the loads + split + LDS stores %d | barrier %d" % (
            med(3) / nk, med(4) / nk, med(5) / max(nk - 1, 1), med(6) / max(nk - 1, 1)))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main()
