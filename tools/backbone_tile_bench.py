"""Does running the ResNet-50 stages on fewer frames at a time (intermediates inside the 256 MB Infinity Cache between producer
and consumer) beat the whole 8-frame batch?  hipGraph replays of both schedules on the bench's input size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomatching_amd import ops  # noqa: E402
from gomatching_amd.config import setup_cfg  # noqa: E402
from gomatching_amd.modeling import ResNet50  # noqa: E402
from gomatching_amd.weights import synth_state_dict  # noqa: E402

DEV = "cuda"
cfg = setup_cfg(builtin="icdar15")
net = ResNet50(synth_state_dict(cfg, seed=0), DEV)
x = torch.randn(8, 1000, 1778, 4, device=DEV)
x[..., 3] = 0


def sched(chunk):
    outs = []
    for b in range(0, 8, chunk):
        outs.append(net.forward(x[b:b + chunk]))
    return outs


def graph_of(chunk):
    sched(chunk)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = sched(chunk)
    return g, keep


def timeit(g, n=10):
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


graphs = {c: graph_of(c) for c in (8, 4, 2, 1)}
for rnd in range(2):
    print("round %d: " % rnd + "  ".join("%d frames at a time %.2f ms" % (c, timeit(graphs[c][0])) for c in (8, 4, 2, 1)))
