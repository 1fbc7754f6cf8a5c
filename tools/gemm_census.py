"""Census of every GEMM / conv launch of one bench step (8 frames, 1000x1778): which kernel family it takes,
its shape, launch count and event-timed duration.  Usage (GPU box): python tools/gemm_census.py"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gomatching_amd import ops  # noqa: E402
from gomatching_amd.config import setup_cfg  # noqa: E402
from gomatching_amd.predictor import new_time_cost, resized_shape  # noqa: E402
from gomatching_amd.synth import make_clip  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = setup_cfg(builtin="icdar15")
    cfg.MODEL.DEVICE = "cuda"
    model, sd = bench.build_model(cfg, dev)
    model.use_graphs = False                                  # per-launch events need eager launches
    h, w = resized_shape(720, 1280, cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
    clip = make_clip(8, h, w, clip_id=0)
    inputs = [{"image": torch.as_tensor(f.astype("float32").transpose(2, 0, 1)).to(dev), "height": 720, "width": 1280}
              for f in clip]
    bench.calibrate(model, inputs)
    model.batch_inference(inputs, 0, 0, [], new_time_cost())
    torch.cuda.synchronize()
    log = []
    real_gemm, real_conv = ops.gemm, ops.conv2d_nhwc

    def timed(kind, shape, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        log.append((kind, shape, e0, e1))
        return out

    def gemm(A, W, *a, **k):
        M = k.get("M") or (k["rows"].numel() if k.get("rows") is not None else A.shape[0])
        split = isinstance(W, ops.SplitWeight)
        N, K = (W.N, W.K) if split else tuple(W.shape)
        fam = "bf16x6" if split else ("fp32-splitk" if (k.get("splitk") or (k.get("splitk") is None and 0 < M <= 128
                                                                               and N * K >= (1 << 18) and K >= 256))
                                      else "fp32")
        return timed(fam, (M, N, K), lambda: real_gemm(A, W, *a, **k))

    def conv(x, w, *a, **k):
        split = isinstance(w, ops.SplitWeight)
        cs = w.conv_shape if split else tuple(w.shape)
        return timed("conv-" + ("bf16x6" if split else "fp32"), (tuple(x.shape), cs, k.get("stride", 1)),
                     lambda: real_conv(x, w, *a, **k))

    ops.gemm, ops.conv2d_nhwc = gemm, conv
    import gomatching_amd.modeling.backbone as bb, gomatching_amd.modeling.deepsolo as ds, \
        gomatching_amd.modeling.roi_heads as rh, gomatching_amd.modeling.meta_arch as ma
    model.batch_inference(inputs, 0, 0, [], new_time_cost())
    torch.cuda.synchronize()
    ops.gemm, ops.conv2d_nhwc = real_gemm, real_conv
    agg = collections.OrderedDict()
    for kind, shape, e0, e1 in log:
        key = (kind, shape)
        c, t = agg.get(key, (0, 0.0))
        agg[key] = (c + 1, t + e0.elapsed_time(e1) * 1e3)
    fam = collections.Counter()
    for (kind, shape), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fam[kind] += t
        print("%-12s %-46s x%-3d %9.1f us total %8.1f us each" % (kind, shape, c, t, t / c))
    print({k: round(v / 1e3, 2) for k, v in fam.items()}, "ms per step (event-timed, serialised)")


if __name__ == "__main__":
    main()
