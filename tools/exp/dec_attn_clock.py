"""Diagnostic build of the decoder attention kernels (csrc/dec_attn.hip): in-kernel cycles of a workgroup's prologue, of the
weight-stage products, of the end-of-stage waits and barriers, and of everything between (attention, conversions, epilogue).
Generated from the product source (s_memtime stamps); the product kernel carries none.
    python tools/exp/dec_attn_clock.py --build   (here)        python tools/exp/dec_attn_clock.py   (GPU box)
STALE: the text anchors below match the kernel source before the RAW form's template and row loader (round 5); --build asserts until they are updated."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "libdec_attn_clock.so")


def build():
    src = open(os.path.join(ROOT, "gomatching_amd", "csrc", "dec_attn.hip")).read()

    def once(s, a, b, n=1):
        assert s.count(a) == n, (a, s.count(a))
        return s.replace(a, b)
    src = once(src, 'template <bool INTER>\n__global__ __launch_bounds__(256, 1) void dec_attn_kernel(const DecArgs p) {',
               '__device__ unsigned long long g_stamp[4096 * 8];\n#define NOW() __builtin_amdgcn_s_memtime()\n'
               '#define DA_STAGE_T(M) { const unsigned long long ta_ = NOW(); DA_STAGE(M) t_prod += NOW() - ta_; }\n'
               'template <bool INTER>\n__global__ __launch_bounds__(256, 1) void dec_attn_kernel(const DecArgs p) {\n'
               '    const unsigned long long t_start = NOW();\n    unsigned long long t_prod = 0, t_wait = 0, t_bar = 0, t_pro = 0, t_epi = 0;\n')
    n_use = src.count('DA_STAGE(DA_MFMA_')
    src = src.replace('DA_STAGE(DA_MFMA_', 'DA_STAGE_T(DA_MFMA_')
    for name, cnt in (("9", None), ("0", None)):
        pass
    src = once(src, '    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");                                                          \\\n    __syncthreads();',
               '    { const unsigned long long tc_ = NOW(); asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); const unsigned long long td_ = NOW(); __syncthreads(); t_wait += td_ - tc_; t_bar += NOW() - td_; } \\\n')
    src = once(src, '    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \\\n    __syncthreads();',
               '    { const unsigned long long tc_ = NOW(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long td_ = NOW(); __syncthreads(); t_wait += td_ - tc_; t_bar += NOW() - td_; } \\\n')
    src = once(src, '    load_rows(false);\n    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    __syncthreads();\n',
               '    load_rows(false);\n    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    __syncthreads();\n    t_pro = NOW() - t_start;\n')
    src = once(src, '    // ---- residual + LayerNorm in registers: lane (token, fh)', '    const unsigned long long t_e0 = NOW();\n    // ---- residual + LayerNorm in registers: lane (token, fh)')
    src = once(src, '    // an operand left fp16\'s range, or a result is not finite',
               '    t_epi = NOW() - t_e0;\n'
               '    if (blockIdx.x < 4096 && lane == 0 && wave == 0) {\n        unsigned long long* o = g_stamp + blockIdx.x * 8;\n'
               '        o[0] = NOW() - t_start; o[1] = t_pro; o[2] = t_prod; o[3] = t_wait; o[4] = t_bar; o[5] = t_epi;\n    }\n'
               '    // an operand left fp16\'s range, or a result is not finite')
    src += ('\nextern "C" int dec_attn_clock_read(unsigned long long* host) {\n'
            '    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 4096 * 8);\n}\n')
    gen = os.path.join(HERE, "_dec_attn_clock_gen.hip")
    open(gen, "w").write(src)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Xclang", "-target-feature", "-Xclang",
                           "-packed-fp32-ops", "-I", os.path.join(ROOT, "gomatching_amd", "csrc"), "-I", os.path.join(ROOT, "include"), gen, "-o", SO])
    os.remove(gen)
    print("built", SO, "(%d timed stages)" % n_use)


def main():
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from gomatching_amd import ops
    so = ctypes.CDLL(SO)
    vp, ci, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
    so.gom_dec_attn_f32.argtypes = [vp, ci, vp, ci, vp, ctypes.c_float, vp, ci, ci, ci, ci, ci, vp, vp]
    so.dec_attn_clock_read.argtypes = [vp]
    dev = "cuda"
    B, nq, P = 8, 100, 25
    Q = B * nq * P
    g = torch.Generator().manual_seed(0)
    in_w = (torch.randn(768, 256, generator=g) / 16).to(dev); in_b = (torch.randn(768, generator=g) * 0.1).to(dev)
    out_w = (torch.randn(256, 256, generator=g) / 16).to(dev); out_b = (torch.randn(256, generator=g) * 0.1).to(dev)
    gamma, beta = (torch.rand(256, generator=g) + 0.5).to(dev), (torch.randn(256, generator=g) * 0.1).to(dev)
    x, pos = torch.randn(Q, 256, generator=g).to(dev), torch.randn(Q, 256, generator=g).to(dev)
    y = torch.empty_like(x)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    host = np.zeros((4096, 8), np.uint64)
    for inter in (0, 1):
        blk = ops.DecAttnBlock(in_w, in_b, out_w, out_b, gamma, beta, bool(inter))
        groups, G, inner = (B * P, nq, P) if inter else (B * nq, P, 1)
        for _ in range(3):
            rc = so.gom_dec_attn_f32(x.data_ptr(), 256, 0 if inter else pos.data_ptr(), 0 if inter else 256, blk.image.data_ptr(), 1e-5,
                                     y.data_ptr(), 256, groups, G, inner, inter, flag.data_ptr(), None)
            assert rc == 0, rc
        torch.cuda.synchronize()
        assert so.dec_attn_clock_read(host.ctypes.data) == 0
        n = groups if inter else (groups + 3) // 4
        h = host[:n].astype(np.float64)
        tot, pro, prod, wait, bar, epi = [h[:, i].mean() for i in range(6)]
        rest = tot - pro - prod - wait - bar - epi
        print("%s: %d workgroups; cycles per workgroup (wave 0; 100 MHz clock x ~21 -> shader cycles): total %.0f = prologue %.0f + "
              "stage products %.0f (32 x %.0f) + end-of-stage waits %.0f + barriers %.0f + epilogue %.0f + rest (attention, "
              "conversions, row reload) %.0f" % ("inter" if inter else "intra", n, tot, pro, prod, prod / 32, wait, bar, epi, rest))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main()
