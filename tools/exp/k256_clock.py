"""Diagnostic build of the row-resident K = 256 GEMM kernel (gemm_k256.hip): cycles of a workgroup's prologue / chunk loop,
and inside the loop of the product, the scale + store block and the end-of-chunk wait + barrier; in-kernel clock.
Generated from the product source (s_memtime / s_memrealtime stamps); the product kernel carries none.
    python tools/exp/k256_clock.py --build   (here)        python tools/exp/k256_clock.py   (GPU box)"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "libk256_clock.so")


def build():
    src = open(os.path.join(ROOT, "gomatching_amd", "csrc", "gemm_k256.hip")).read()

    def once(s, a, b, n=1):
        assert s.count(a) == n, a
        return s.replace(a, b)
    src = once(src, 'template <bool LINES>\n__global__ __launch_bounds__(256, 2) void gemm_k256_kernel(const RowArgs p) {',
               '__device__ unsigned long long g_stamp[8192 * 12];\n'
               '#define NOW() __builtin_amdgcn_s_memtime()\n'
               'template <bool LINES>\n__global__ __launch_bounds__(256, 2) void gemm_k256_kernel(const RowArgs p) {\n'
               '    const unsigned long long t_start = NOW(), r_start = __builtin_amdgcn_s_memrealtime();\n'
               '    unsigned long long t_prod = 0, t_store = 0, t_wait = 0, t_bar = 0;\n')
    src = once(src, '    {\n        const float lo = p.relu ? 0.f : -INFINITY;',
               '    const unsigned long long t_loop = NOW(), r_loop = __builtin_amdgcn_s_memrealtime();\n    {\n        const float lo = p.relu ? 0.f : -INFINITY;')
    # both store forms of the kernel (template <bool LINES>): stamps around the product, the scale + store block and the wait
    src = once(src, '                chunk_product(smem + st * CHUNK_BYTES + lane * 16, acc, nsrc, ndst);',
               '                const unsigned long long ta_ = FINE ? NOW() : 0;\n'
               '                chunk_product(smem + st * CHUNK_BYTES + lane * 16, acc, nsrc, ndst);\n'
               '                const unsigned long long tb_ = FINE ? NOW() : 0;', 2)
    for n_ in ("4", "16"):
        src = once(src, '                asm volatile("s_waitcnt vmcnt(%s)" ::: "memory");' % n_,
                   '                const unsigned long long tc_ = FINE ? NOW() : 0;\n'
                   '                asm volatile("s_waitcnt vmcnt(%s)" ::: "memory");\n'
                   '                if (FINE) { t_prod += tb_ - ta_; t_store += tc_ - tb_; t_wait += NOW() - tc_; }' % n_)
    src = once(src, '            __syncthreads();                                 // next stage complete for everybody; nobody still reads this one',
               '            const unsigned long long td_ = FINE ? NOW() : 0;\n            __syncthreads();\n            if (FINE) t_bar += NOW() - td_;')
    src = once(src, '    if (bad && p.flag) atomicOr(p.flag, 1);                  // an operand left fp16',
               '    if (blockIdx.y == 0 && blockIdx.x < 8192 && lane == 0 && wave == 0) {\n'
               '        unsigned long long* o = g_stamp + blockIdx.x * 12;\n'
               '        o[0] = t_start; o[1] = r_start; o[2] = t_loop; o[3] = r_loop; o[4] = NOW(); o[5] = __builtin_amdgcn_s_memrealtime();\n'
               '        o[6] = t_prod; o[7] = t_store; o[8] = t_wait; o[9] = t_bar;\n    }\n'
               '    if (bad && p.flag) atomicOr(p.flag, 1);                  // an operand left fp16')
    src += ('\nextern "C" int k256_clock_read(unsigned long long* host) {\n'
            '    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 8192 * 12);\n}\n')
    src = once(src, '                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_c,',
               '                    if (!NOSTORE || p.M < 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_c,')
    src = once(src, '                    *reinterpret_cast<f32x4*>(p.C + (size_t)row * p.ldc + CW * c + 8 * q + 4 * fh) = v;',
               '                    if (!NOSTORE || p.M < 0) *reinterpret_cast<f32x4*>(p.C + (size_t)row * p.ldc + CW * c + 8 * q + 4 * fh) = v;')
    gen = os.path.join(HERE, "_k256_clock_gen.hip")
    open(gen, "w").write(src)
    for fine in (0, 1, 2):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
                               "-DFINE=%d" % (fine & 1), "-DNOSTORE=%d" % (fine >> 1), "-I", os.path.join(ROOT, "gomatching_amd", "csrc"), "-I", os.path.join(ROOT, "include"), gen, "-o",
                               SO.replace(".so", ("_fine.so", "_nostore.so")[fine - 1]) if fine else SO])
    os.remove(gen)


def main(fine):
    import time
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from gomatching_amd import ops
    so = ctypes.CDLL(SO.replace(".so", ("_fine.so", "_nostore.so")[fine - 1]) if fine else SO)
    vp = ctypes.c_void_p
    so.gom_gemm_k256_f32.argtypes = [vp, vp, ctypes.c_int, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, vp, vp]
    so.k256_clock_read.argtypes = [vp]
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    for M, N in ((297368, 640), (297368, 1536), (20000, 256)):
        w = (torch.randn((N, 256), generator=g) * 0.05).to(dev); b = torch.randn((N,), generator=g).to(dev) * 0.1
        lin = ops.K256Linear(ops.prep_weight(w), b)
        x = torch.randn((M, 256), generator=g).to(dev)
        y = torch.empty((M, N), device=dev)
        ops.K256_MAX_ROWS = 1 << 30
        ref = ops.linear(x, lin, groups=1)
        flag = torch.zeros((1,), dtype=torch.int32, device=dev)

        def run():
            rc = so.gom_gemm_k256_f32(x.data_ptr(), None, 256, lin.image.data_ptr(), None, 0, 0, 0, y.data_ptr(), N, M, N, 256, 1, flag.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        run(); torch.cuda.synchronize()
        assert fine == 2 or torch.equal(y, ref), "stamped build differs from the product kernel"
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
        buf = np.zeros((8192, 12), dtype=np.uint64)
        assert so.k256_clock_read(buf.ctypes.data) == 0
        n = min(8192, (M + 127) // 128)
        s = buf[:n].astype(np.int64)
        chunks = N // 32
        loop_c = s[:, 4] - s[:, 2]
        clk = np.median(loop_c / np.maximum(s[:, 5] - s[:, 3], 1)) * 100.0
        print("M %d N %d%s: launch %.1f us; in-kernel clock %.0f MHz; per workgroup (median of %d): prologue %d | loop %d = %d per chunk (MFMA 48 x 32 = 1536 per wave, two waves per SIMD) | lifetime %d cycles = %.1f us" % (
            M, N, ("", " FINE", " NOSTORE")[fine], us, clk, n, np.median(s[:, 2] - s[:, 0]), np.median(loop_c), np.median(loop_c) / chunks,
            np.median(s[:, 4] - s[:, 0]), np.median(s[:, 4] - s[:, 0]) / clk))
        if fine == 1:
            print("      per chunk (wave 0): product %d | scale + store issue %d | vmcnt wait %d | barrier %d" % tuple(np.median(s[:, 6 + i]) / chunks for i in range(4)))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main(0)
        main(1)
        main(2)                 # the same kernel with its output stores removed: how much of the time is the store path?
