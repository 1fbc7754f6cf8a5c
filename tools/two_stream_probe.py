"""Would the detector of a step run faster as TWO concurrent half-batches (4 + 4 frames on two streams inside one hipGraph)
than as one batch of 8?  Frames are independent everywhere in the detector; half-size launches fill each other's tail rounds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gomatching_amd.config import setup_cfg  # noqa: E402
from gomatching_amd.predictor import new_time_cost  # noqa: E402

dev = torch.device("cuda", 0)
cfg = setup_cfg(builtin="icdar15")
cfg.MODEL.DEVICE = "cuda"
model, sd = bench.build_model(cfg, dev)
x8 = torch.rand(8, 3, 1000, 1778, device=dev) * 255
tc = new_time_cost()


def core(x):
    return model._detect_core(x, ("f32", (1000, 1778), None), {k: 0.0 for k in tc if k != "_sync"})


def one():
    return core(x8)


streams = [torch.cuda.Stream() for _ in range(8)]


def split(parts):
    def fn():
        cur = torch.cuda.current_stream()
        outs, o = [], 0
        for i, n in enumerate(parts):
            st = streams[i]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(core(x8[o:o + n]))
            o += n
        for i in range(len(parts)):
            cur.wait_stream(streams[i])
        return outs
    return fn


def graph_of(fn):
    fn()
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fn()
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


cases = {"8": one, "4+4": split([4, 4]), "2+2+2+2": split([2, 2, 2, 2]), "3+3+2": split([3, 3, 2]), "5+3": split([5, 3]),
         "1 x 8": split([1] * 8)}
graphs = {k: graph_of(f) for k, f in cases.items()}
for rnd in range(2):
    print("round %d: " % rnd + "  ".join("%s: %.2f ms" % (k, timeit(g[0])) for k, g in graphs.items()))
