"""Fused FFN block against the three-launch path (GEMM, GEMM, LayerNorm), interleaved in one process.
    python tools/ffn_bench.py [M ...]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from gomatching_amd import ops

dev = "cuda"
ops.GEMM_MODE = "f16x3"
g = torch.Generator().manual_seed(0)
F = 1024
w1 = (torch.randn((F, 256), generator=g) * 0.05).to(dev); b1 = torch.randn((F,), generator=g).to(dev) * 0.1
w2 = (torch.randn((256, F), generator=g) * 0.05).to(dev); b2 = torch.randn((256,), generator=g).to(dev) * 0.1
ga = torch.ones((256,), device=dev); be = torch.zeros((256,), device=dev)
ffn = ops.FusedFFN(w1, b1, w2, b2, ga, be)
p1, p2 = ops.prep_weight(w1), ops.prep_weight(w2)
for M in [int(a) for a in sys.argv[1:]] or [297368, 20000]:
    x = torch.randn((M, 256), generator=g).to(dev)
    y = torch.empty_like(x)

    def fused():
        ops.ffn_fused_ln(x, ffn, out=y)

    def three():
        h = ops.gemm(x, p1, bias=b1, relu=True)
        z = ops.gemm(h, p2, bias=b2, R=x)
        ops.layernorm(z, ga, be, out=y)

    res = {}
    for name, fn in (("fused", fused), ("three", three)):
        for _ in range(3):
            fn()
    for rnd in range(5):
        for name, fn in (("fused", fused), ("three", three)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 5 * 1e3)
    fl = 4.0 * M * 256 * F
    for name in res:
        us = sorted(res[name])[len(res[name]) // 2]
        print("M %7d %-6s median %8.1f us  min %8.1f us  %6.1f TFLOP/s fp32-equivalent (%.3f of 833)" % (
            M, name, us, min(res[name]), fl / us / 1e6, fl / us / 1e6 / 833.3), flush=True)
