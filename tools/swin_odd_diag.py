"""Diagnostic: run-to-run determinism of the Swin e2e path at 90x130 (bitwise comparison of stage outputs between runs)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from helpers import mini_cfg
from gomatching_amd import ops
from gomatching_amd.modeling import GoMatching
from gomatching_amd.synth import make_clip
from gomatching_amd.weights import synth_state_dict

DEV = "cuda"
hw = (90, 130)
cfg = mini_cfg("icdar15", device=DEV)
cfg.MODEL.BACKBONE.NAME = "build_swin_backbone"
sd = synth_state_dict(cfg, seed=7, cls_bias={"detection_transformer.ctrl_point_class.0.bias": 0.8,
                                              "roi_heads.rescoring_head.bias": 0.8})
clip = make_clip(6, hw[0], hw[1], clip_id=2)
images = [torch.as_tensor(f.astype("float32").transpose(2, 0, 1)) for f in clip]
junk = [torch.randn(1 << 22, device=DEV) * 1e3 for _ in range(8)]
for mode in ("bf16x6", "f16x3"):
    ops.GEMM_MODE = mode
    first = None
    for rep in range(12):
        del junk
        junk = [torch.randn(1 << (18 + rep % 5), device=DEV) * 1e3 for _ in range(8)]   # dirty the allocator
        model = GoMatching(cfg, sd, device=DEV, frames_per_step=3)
        model.use_graphs = rep % 2 == 0
        x, _ = model.preprocess_image([{"image": im} for im in images[:3]])
        feats = model.backbone.forward(x)
        taps = {}
        out = model.detection_transformer.forward([feats[k] for k in model.feature_names], taps=taps)
        tc = {k: 0.0 for k in ("pre_process", "backbone", "detector", "rescore", "tracker", "short_match", "long_match",
                               "post_process", "total_time")}
        insts, idc = model.batch_inference([{"image": im, "height": hw[0], "width": hw[1]} for im in images], 0, 0, [], tc)
        cur = {("feat", k): v.clone() for k, v in feats.items()}
        cur.update({("tap", k): v.clone() for k, v in taps.items() if torch.is_tensor(v)})
        cur.update({("out", k): v.clone() for k, v in out.items() if torch.is_tensor(v)})
        cur[("ids",)] = torch.cat([i.track_ids.reshape(-1).float() for i in insts])
        cur[("scores",)] = torch.cat([i.scores.reshape(-1) for i in insts])
        if first is None:
            first = cur
        else:
            diffs = [(k, float((first[k].float() - v.float()).abs().max())) for k, v in cur.items()
                     if first[k].shape != v.shape or not torch.equal(first[k], v)]
            print(mode, rep, "graphs" if model.use_graphs else "eager", "id_count", int(idc), "DIFF" if diffs else "same", diffs[:6])
