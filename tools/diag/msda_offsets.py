"""Diagnostic (GPU box): how far from the query's own (projected) pixel do the encoder's MSDA samples fall?  The raw sampling
offsets of a deformable-attention layer are in PIXELS of the sampled level (loc = ref + off / (W_l, H_l)), so the distribution of
|off| decides whether a per-workgroup LDS window of the value map (query tile + halo of r pixels per level) could serve the
gather.  Prints, per encoder layer of the bench workload (8 frames 1000x1778, synthetic weights), quantiles of max(|dx|, |dy|)
and the fraction of samples (plain and attention-weighted) inside r = 2, 3, 4, 6, 8, 12, 16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gomatching_amd import ops                                   # noqa: E402
from gomatching_amd.config import setup_cfg                      # noqa: E402
from gomatching_amd.modeling import GoMatching                   # noqa: E402
from gomatching_amd.predictor import GoMBatchPredictor           # noqa: E402
from gomatching_amd.synth import make_clip                       # noqa: E402
from gomatching_amd.weights import synth_state_dict              # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
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
for li, L in enumerate(det.enc):
    rv = ops.linear(src, L["attn"]["raw_value"], R=geo["pos_w"][li], r_cols=384, r_period=S if geo["pos_periodic"] else 0)
    off = rv[:, :256].view(B * S, 8, 4, 4, 2)                    # head, level, point, (x, y): pixels of the sampled level
    w = torch.softmax(rv[:, 256:384].view(B * S, 8, 16), -1).view(B * S, 8, 4, 4)
    r = off.abs().max(-1)[0]                                     # Chebyshev distance of the sample from the projected pixel
    qs = torch.quantile(r.reshape(-1)[:: max(1, r.numel() // 4000000)].float(), torch.tensor([0.5, 0.9, 0.99, 0.999], device=r.device))
    line = "layer %d: |off| median %.2f  p90 %.2f  p99 %.2f  p99.9 %.2f  max %.1f |" % (li, *[float(v) for v in qs], float(r.max()))
    for rad in (2, 3, 4, 6, 8, 12, 16):
        inside = (r <= rad).float()
        line += "  r<=%d: %.4f (w %.4f)" % (rad, float(inside.mean()), float((inside * w).sum() / w.sum()))
    print(line, flush=True)
    samp = ops.msda_fused(rv, geo["enc_ref"], rv[:, 384:], S * 640, geo["shapes"], geo["lsi"], B, S, geo["vr"])
    src = ops.proj_ln(samp, L["out_ln"], src)
    src = ops.ffn_fused_ln(src, L["ffn"])
