"""out_proj + LayerNorm kernel (csrc/proj_ln.hip) at the encoder's row count: with residual, without, dot form, and the launches
the dot form replaces (tile GEMM + LayerNorm + N = 1 GEMM).    python tools/proj_ln_bench.py [M ...]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from gomatching_amd import ops

dev = "cuda"
ops.GEMM_MODE = "f16x3"
g = torch.Generator().manual_seed(0)
w = (torch.randn((256, 256), generator=g) * 0.06).to(dev)
b = torch.randn((256,), generator=g).to(dev) * 0.1
ga, be = torch.ones((256,), device=dev), torch.zeros((256,), device=dev)
sw = ops.split_weight(w, kind="f16x3")
blk = ops.ProjLN(sw, b, ga, be)
cw = torch.randn((1, 256), generator=g).to(dev) * 0.1
cb = torch.zeros((1,), device=dev)
from gomatching_amd import lib
Lh = lib.load()
for M in [int(a) for a in sys.argv[1:]] or [297368, 20000]:
    for v2 in (0, 1, 0, 1):
        ops.PROJ_LN_V2 = bool(v2)
        print("== %s form" % ("two-workgroups-per-CU (64-row tiles)" if v2 else "128-row"), flush=True)
        x = torch.randn((M, 256), generator=g).to(dev)
        r = torch.randn((M, 256), generator=g).to(dev)
        y = torch.empty_like(x)
        cases = {
            "proj_ln with residual": lambda: ops.proj_ln(x, blk, r, out=y),
            "proj_ln, no residual": lambda: ops.proj_ln(x, blk, None, out=y),
            "proj_ln dot form": lambda: ops.proj_ln_dot(x, blk, cw.view(256), 0.0),
            "GEMM + LayerNorm + N=1 GEMM": lambda: ops.gemm(ops.layernorm(ops.gemm(x, sw, bias=b), ga, be), cw, bias=cb),
        }
        for name, fn in cases.items():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(15):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            print("M %7d  %-30s median %8.1f us  min %8.1f us" % (M, name, ts[len(ts) // 2], ts[0]))
