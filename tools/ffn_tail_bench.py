import os, sys
sys.path.insert(0, os.getcwd())
import torch
from gomatching_amd import ops, lib
dev = "cuda"
g = torch.Generator().manual_seed(0)
F = 1024
w1 = (torch.randn((F, 256), generator=g) * 0.05).to(dev); b1 = torch.randn((F,), generator=g).to(dev) * 0.1
w2 = (torch.randn((256, F), generator=g) * 0.05).to(dev); b2 = torch.randn((256,), generator=g).to(dev) * 0.1
ga = torch.ones((256,), device=dev); be = torch.zeros((256,), device=dev)
ffn = ops.FusedFFN(w1, b1, w2, b2, ga, be)
L = lib.load()
for M in (297368, 485120):
    x = torch.randn((M, 256), generator=g).to(dev); y = torch.empty_like(x)
    def run(mode):
        L.gom_ffn_set_half_tail(mode); ops.ffn_fused_ln(x, ffn, out=y)
    for m in (0, 1): run(m)
    for rnd in range(4):
        out = []
        for mode in (0, 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): run(mode)
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 8 * 1e3)
        print("M %d round %d: one launch %.1f us | half-height tail %.1f us" % (M, rnd, out[0], out[1]), flush=True)
L.gom_ffn_set_half_tail(1)
