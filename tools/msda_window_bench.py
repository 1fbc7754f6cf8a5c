"""Interleaved A/B on one GPU: the encoder's MSDA call at the bench's shape (8 frames x 37 171 tokens), level-0 queries served
from LDS windows (csrc/msda.hip msda_window_kernel) against everything on the lane-distributed kernel.  Inputs: the bench
model's own first encoder layer (offsets as the synthetic weights produce them)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomatching_amd import lib, ops                              # noqa: E402
from gomatching_amd.config import setup_cfg                      # noqa: E402
from gomatching_amd.modeling import GoMatching                   # noqa: E402
from gomatching_amd.predictor import GoMBatchPredictor           # noqa: E402
from gomatching_amd.synth import make_clip                       # noqa: E402
from gomatching_amd.weights import synth_state_dict              # noqa: E402

B = 8
cfg = setup_cfg(builtin="icdar15")
cfg.MODEL.DEVICE = "cuda"
clip = make_clip(B, 720, 1280, clip_id=0, num_rects=12)
inputs, _ = GoMBatchPredictor(cfg, None).prepare([f[:, :, ::-1] for f in clip])
model = GoMatching(cfg, synth_state_dict(cfg, seed=0), device="cuda", frames_per_step=B, use_graphs=False)
x, _ = model.preprocess_image(inputs)
feats = model.backbone.forward(x)
det = model.detection_transformer
src, geo = det.input_tokens([feats[k] for k in model.feature_names], B)
S = geo["S"]
L0 = det.enc[0]
rv = ops.linear(src, L0["attn"]["raw_value"], R=geo["pos_w"][0], r_cols=384, r_period=S if geo["pos_periodic"] else 0)
Lh = lib.load()


def run(window, overlap=1):
    Lh.gom_msda_set_overlap(overlap)
    Lh.gom_msda_set_window(int(window))
    return ops.msda_fused(rv, geo["enc_ref"], rv[:, 384:], S * 640, geo["shapes"], geo["lsi"], B, S, None, encoder_hw0=geo["hw0"])


a, b, c, d = run(0), run(1), run(2), run(3)
import hashlib
print("sha1 lane %s window %s" % (hashlib.sha1(a.cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha1(b.cpu().numpy().tobytes()).hexdigest()[:16]))
print("bit-identical:", bool(torch.equal(a, b)), bool(torch.equal(a, c)), bool(torch.equal(a, d)), "max |d| %.3e" % float((a - b).abs().max()))


def burst(window, n=10, overlap=1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run(window, overlap)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for rnd in range(4):
    print("round %d: lane-distributed kernel %.1f us | level-0 from LDS windows + coarser levels on the lane kernel: single buffer %.1f us, "
          "double-buffered %.1f us, single buffer, TWO corner addresses computed at the owner %.1f us" % (rnd, burst(0), burst(1), burst(2), burst(3)), flush=True)
Lh.gom_msda_set_window(1)
