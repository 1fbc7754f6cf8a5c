"""Shared by the CPU (oracle) and GPU (HIP) tests of the full-size reference fixtures tests/golden/full_c{1,2}.npz
(produced by oracle/gen_golden_full.py from the reference's own DeepSolo module): the case's frame, its weights, and the
comparison rule."""
import hashlib

import numpy as np
import torch

from helpers import golden
from gomatching_amd.config import setup_cfg
from gomatching_amd.weights import synth_state_dict

SRC_HW = {"c1": ((640, 640), False), "c2": ((720, 1280), True)}


def case(name, device="cpu"):
    g = golden("full_%s.npz" % name)
    cfg = setup_cfg(builtin="icdar15")
    cfg.MODEL.DEVICE = device
    sd = synth_state_dict(cfg, seed=int(g["seed"][0]),
                          cls_bias={"detection_transformer.ctrl_point_class.0.bias": float(g["cls_bias"][0])})
    from gomatching_amd.synth import make_clip
    (h, w), resize = SRC_HW[name]
    clip = make_clip(1, h, w, clip_id=0, num_rects=12)
    if resize:
        from gomatching_amd.predictor import GoMBatchPredictor
        pcfg = setup_cfg(builtin="icdar15")
        image = GoMBatchPredictor(pcfg, None).prepare([clip[0][:, :, ::-1]])[0][0]["image"].contiguous()
    else:
        image = torch.as_tensor(clip[0].astype("float32").transpose(2, 0, 1)).contiguous()
    assert tuple(image.shape[-2:]) == tuple(int(v) for v in g["hw"])
    # the frame the reference saw, bit for bit (seeded generator + Pillow's fixed-point resize)
    assert hashlib.sha1(image.numpy().tobytes()).digest() == g["image_sha1"].tobytes(), "synthetic frame differs from the fixture's"
    return g, cfg, sd, image


def compare(g, out, topk, nq=100, P=25, tol=2e-4, what=""):
    """out: the five outputs as tensors reshapeable to [nq, P, C]; topk: [nq] token indices in rank order.
    The reference's winners must be reproduced as a SET; two winners may swap RANK only where the reference's own logits are
    closer than 2e-5 (its fp32 noise level at S = 37 171: gom_golden_full prints the gaps), and the queries of swapped ranks are
    then excluded from the per-query comparison (a query slot = a learned point embedding + the token of that rank)."""
    ref_idx = g["topk_idx"].astype(np.int64)
    got_idx = np.asarray(topk.cpu() if hasattr(topk, "cpu") else topk).reshape(-1).astype(np.int64)
    assert sorted(ref_idx.tolist()) == sorted(got_idx.tolist()), "%s: top-k token SET differs from the reference's" % what
    moved = np.nonzero(ref_idx != got_idx)[0]
    val = g["topk_val"]
    for i in moved:
        j = int(np.nonzero(ref_idx == got_idx[i])[0][0])
        assert abs(float(val[i]) - float(val[j])) < 2e-5, "%s: rank %d <-> %d swapped across a gap of %.2e" % (
            what, i, j, abs(float(val[i]) - float(val[j])))
    keep = np.ones((nq,), bool)
    keep[moved] = False
    assert keep.sum() >= nq - 6, "%s: %d ranks moved" % (what, len(moved))
    v = {k: out[k].detach().cpu().float().reshape(nq, P, -1) for k in ("pred_logits", "pred_text_logits", "pred_ctrl_points",
                                                                        "pred_bd_points", "query_features")}
    err = {}
    err["logit_mean"] = np.abs(v["pred_logits"].mean(1).reshape(nq).numpy() - g["logit_mean"])[keep].max()
    err["ctrl"] = np.abs(v["pred_ctrl_points"].numpy() - g["ctrl"])[keep].max()
    err["bd"] = np.abs(v["pred_bd_points"].numpy() - g["bd"])[keep].max()
    q = [int(x) for x in g["queries"] if keep[int(x)]]
    sel = [i for i, x in enumerate(g["queries"]) if keep[int(x)]]
    err["q_logits"] = np.abs(v["pred_logits"][q].numpy() - g["q_logits"][sel]).max()
    err["q_text"] = np.abs(v["pred_text_logits"][q].numpy() - g["q_text"][sel]).max()
    err["q_feat"] = np.abs(v["query_features"][q].numpy() - g["q_feat"][sel]).max()
    for k, e in err.items():
        assert e <= tol, "%s: %s max|d| = %.3e (tol %.1e)" % (what, k, e, tol)
    # characters: identical wherever the reference's own top-2 logit gap is above the tolerance
    recs = v["pred_text_logits"].argmax(-1).numpy()
    sure = (g["text_top2_gap"] > 4 * tol) & keep[:, None]
    assert (recs == g["recs"])[sure].all(), "%s: character arg-max differs" % what
    return err, len(moved)
