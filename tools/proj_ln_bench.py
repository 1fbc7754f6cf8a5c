"""out_proj + residual + LayerNorm as one launch (proj_ln.hip) against GEMM (residual in the epilogue) + LayerNorm, alternating
bursts, encoder (M = 297 368) and decoder (M = 20 000) sizes."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from gomatching_amd import ops

ops.GEMM_MODE = "f16x3"
dev = "cuda"


def burst(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


g = torch.Generator().manual_seed(0)
for M in (297368, 20000):
    x = torch.randn((M, 256), generator=g).to(dev)
    r = torch.randn((M, 256), generator=g).to(dev)
    w = (torch.randn((256, 256), generator=g) * 0.06).to(dev)
    b = torch.randn((256,), generator=g).to(dev)
    ga, be = torch.ones((256,), device=dev), torch.zeros((256,), device=dev)
    sw = ops.split_weight(w, kind="f16x3")
    blk = ops.ProjLN(sw, b, ga, be)
    lin = ops.K256Linear(sw, b)
    y, z = torch.empty_like(x), torch.empty_like(x)
    one = lambda: ops.proj_ln(x, blk, r, out=y)
    two = lambda: ops.layernorm(ops.linear(x, lin, R=r, out=z), ga, be, out=y)
    for f in (one, two):
        f()
    torch.cuda.synchronize()
    a, c = [], []
    for _ in range(7):
        a.append(burst(one))
        c.append(burst(two))
    print("M %6d: one launch %7.1f us | GEMM + LayerNorm %7.1f us" % (M, sorted(a)[3], sorted(c)[3]), flush=True)
