"""A/B: the fused FFN block with its LDS-staged epilogue (csrc/ffn_fused.hip) against the same chunk pipeline with the
register epilogue of csrc/dec_tail.hip (residual / LayerNorm / stores in the accumulator layout), encoder- and decoder-sized."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gomatching_amd import lib, ops                              # noqa: E402
from test_ffn_gpu import _case                                   # noqa: E402

L = lib.load()
for M in (297368, 20000):
    x, w1, b1, w2, b2, ga, be = [v.to("cuda") for v in _case(M, 1024, seed=3)]
    ffn = ops.FusedFFN(w1, b1, w2, b2, ga, be)
    y0 = ops.ffn_fused_ln(x, ffn)
    y1 = torch.empty_like(y0)

    def reg():
        ops.check(L.gom_ffn_fused_ln_reg_f32(ops._p(x), 256, ops._p(ffn.image), ops._p(ffn.inv2), ops._p(ffn.b2), ops._p(ffn.gamma),
                                             ops._p(ffn.beta), ffn.eps, ops._p(y1), 256, M, 256, 1024, ops._p(ops.range_flag(x.device)),
                                             ops._stream()), "gom_ffn_fused_ln_reg_f32")

    def lds():
        ops.ffn_fused_ln(x, ffn, out=y0)

    reg()
    torch.cuda.synchronize()
    print("M = %d: max |reg - lds| %.2e" % (M, float((y0 - y1).abs().max())))

    def burst(fn, n=10):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    for rnd in range(4):
        print("  round %d: LDS-staged epilogue %.1f us | register epilogue %.1f us" % (rnd, burst(lds), burst(reg)), flush=True)
