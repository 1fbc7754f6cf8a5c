"""Time the fused stem launch (conv 7x7 / 2 + BN + ReLU + max-pool) on the bench's 8 frames of 1000 x 1778."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomatching_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)
x = torch.randn(8, 1000, 1778, 4, generator=g).to(dev)
x[..., 3] = 0
w = (torch.randn(64, 7, 7, 4, generator=g) / 14).to(dev)
w[..., 3] = 0
sw = ops.prep_conv_weight(w)
sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
for _ in range(3):
    y = ops.stem_conv_pool(x, sw, scale=sc, shift=sh)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.stem_conv_pool(x, sw, scale=sc, shift=sh)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print("stem_conv_pool: median %.1f us (min %.1f)" % (sorted(ts)[5], min(ts)))
