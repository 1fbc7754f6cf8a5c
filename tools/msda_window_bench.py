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
import ctypes                                                     # noqa: E402
import hashlib                                                    # noqa: E402


def run(mask, rv_=None):
    """mask 0: everything on the lane-distributed (gather) kernel; 1: level-0 queries from LDS windows; 3: level-0 and level-1."""
    ops.MSDA_WINDOW = mask != 0
    ops.MSDA_WINDOW_L1 = mask == 3
    r = rv if rv_ is None else rv_
    return ops.msda_fused(r, geo["enc_ref"], r[:, 384:], S * 640, geo["shapes"], geo["lsi"], B, S, None, encoder_hw0=geo["hw0"])


def burst(mask, n=10, rv_=None):
    run(mask, rv_)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run(mask, rv_)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


a, b, c = run(0), run(1), run(3)
print("sha1 gather %s windows %s" % (hashlib.sha1(a.cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha1(c.cpu().numpy().tobytes()).hexdigest()[:16]))
print("bit-identical:", bool(torch.equal(a, b)), bool(torch.equal(a, c)), "max |d| %.3e" % float((a - c).abs().max()))
for rnd in range(4):
    print("round %d: gather kernel %.1f us | level-0 windows + gather for the rest %.1f us | level-0 + level-1 windows + gather for levels 2-3 %.1f us"
          % (rnd, burst(0), burst(1), burst(3)), flush=True)

# ---- what the window assumption is worth when the offsets are larger than the synthetic weights make them: the
#      sampling offsets (raw columns [0, 256), in pixels of the sampled level) scaled x1 / x2 / x4; halo R = 5 pixels ----
counter = torch.zeros((1,), dtype=torch.int32, device="cuda")
Lh.gom_msda_window_count_fallbacks(ctypes.c_void_p(counter.data_ptr()))
h0, w0, h1, w1 = geo["hw0"]
groups = B * 8 * (-(-h0 // 8) * -(-w0 // 16) * 16 + -(-h1 // 8) * -(-w1 // 16) * 16)
off = rv[:, :256].float()
print("offsets of the bench model's first encoder layer: |off| mean %.2f, p99 %.2f, max %.2f pixels"
      % (float(off.abs().mean()), float(off.abs().flatten()[::97].quantile(0.99)), float(off.abs().max())))
for scale in (1.0, 2.0, 4.0):
    r2 = rv.clone()
    r2[:, :256] *= scale
    counter.zero_()
    ref_out = run(0, r2)
    got = run(3, r2)
    torch.cuda.synchronize()
    slow = int(counter.item())
    print("offsets x %.0f: %6.2f %% of the octet groups fall back to the gather path (%d of %d), bit-identical %s | gather %.1f us, "
          "level-0 windows %.1f us, level-0 + level-1 windows %.1f us"
          % (scale, 100.0 * slow / groups, slow, groups, bool(torch.equal(ref_out, got)), burst(0, 10, r2), burst(1, 10, r2), burst(3, 10, r2)), flush=True)
Lh.gom_msda_window_count_fallbacks(None)
