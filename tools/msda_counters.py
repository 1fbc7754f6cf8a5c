"""Encoder-sized fused MSDA launches (8 x 37 171 queries, offsets of a few pixels) for the counter passes of
tools/msda_counters.sh; `summarise` turns rocprofv3's counter_collection.csv files into per-launch figures."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(amp=3.0, launches=4):
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
    raw = torch.randn((B * S, 384), generator=g).to(dev)
    raw[:, :256] *= amp / 1.7
    ts = []
    for _ in range(launches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.msda_fused(raw, ref, value[:, 384:], S * 640, ss.to(dev), lsi.to(dev), B, S)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print("msda encoder-sized launch: %s us" % ["%.0f" % t for t in ts])


def summarise(paths):
    tot = {}
    for path in paths:
        with open(path) as f:
            for row in csv.DictReader(f):
                if "msda" not in row["Kernel_Name"]:
                    continue
                d = tot.setdefault(row["Counter_Name"], [0, 0.0])
                d[0] += 1
                d[1] += float(row["Counter_Value"])
    for k in sorted(tot):
        n, v = tot[k]
        print("%-40s per launch %.4g (%d records)" % (k, v / n, n))
    return {k: v[1] / v[0] for k, v in tot.items()}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "summarise":
        summarise(sys.argv[2:])
    else:
        run(float(sys.argv[1]) if len(sys.argv) > 1 else 3.0)
