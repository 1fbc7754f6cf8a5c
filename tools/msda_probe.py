"""What bounds the fused MSDA kernel?  Encoder-sized call (8 x 37 171 queries) with (a) random offsets of a few pixels,
(b) ALL offsets zero (every point of a query samples the same 4 pixels per level: same instruction count, perfect L1 hits),
(c) offsets of +-40 pixels (poor locality)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from gomatching_amd import ops

dev = "cuda"
shapes = [(125, 223), (63, 112), (32, 56), (16, 28)]
ss = torch.as_tensor(shapes, dtype=torch.long)
lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
S, B = int(ss.prod(1).sum()), 8
g = torch.Generator().manual_seed(0)
value = torch.randn((B * S, 640), generator=g).to(dev)
ref = ops.broadcast_rows(ops.encoder_reference_points(ss.to(dev), lsi.to(dev), S, None), B).view(B * S, 1, 2)
for name, amp in (("offsets ~ +-3 px", 3.0), ("offsets = 0", 0.0), ("offsets ~ +-40 px", 40.0)):
    raw = torch.randn((B * S, 384), generator=g).to(dev)
    raw[:, :256] *= amp / 1.7
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.msda_fused(raw, ref, value[:, 384:], S * 640, ss.to(dev), lsi.to(dev), B, S)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    from gomatching_amd import lib
    a = ops.msda_fused(raw, ref, value[:, 384:], S * 640, ss.to(dev), lsi.to(dev), B, S)
    lib.load().gom_msda_set_lane_distributed(0)
    t0 = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b_ = ops.msda_fused(raw, ref, value[:, 384:], S * 640, ss.to(dev), lsi.to(dev), B, S)
        e1.record()
        torch.cuda.synchronize()
        t0.append(e0.elapsed_time(e1) * 1e3)
    lib.load().gom_msda_set_lane_distributed(1)
    print("%-20s lane-distributed %8.1f us (min %.1f) | one lane does all %8.1f us (min %.1f) | max |d| %.2e, identical: %s" % (
        name, sorted(ts)[len(ts) // 2], min(ts), sorted(t0)[len(t0) // 2], min(t0), float((a - b_).abs().max()), bool(torch.equal(a, b_))), flush=True)
