"""Full-size golden fixtures from the REFERENCE's own DeepSolo module (SURVEY.md §8-c item iii; container-only).

TEST INFRASTRUCTURE.  Run in the build container, where /root/reference exists:

    python -m oracle.gen_golden_full            # writes tests/golden/full_c1.npz, full_c2.npz

The mini-geometry fixtures of oracle/gen_golden.py cannot show the one hazard that only exists at full size: the
proposal stage's top-k over S = 8 500 (C1: 640x640 fed directly) or S = 37 171 (C2: 1280x720 -> 1000x1778) class logits
(third_party/adet/layers/deformable_transformer.py:183-199), where near-ties decide which tokens become queries.  Here the
reference's unmodified DETECTION_TRANSFORMER_WOBACKBONE (third_party/adet/modeling/model/detection_transformer_wobackbone.py:
159-270) runs, through oracle/ref_shim.py, on ONE full-size synthetic frame with the repo's synthetic weights and 100
queries; its inputs are the res3/res4/res5 maps of the oracle's R-50 restatement (Detectron2's backbone is absent from
/root/reference).  Stored (small, numeric only): the top-k token indices the reference's own torch.topk call returned,
SHA-1 digests of its five output tensors (regeneration check), and, for parity at tolerance, the per-query point means of
the class logits, all control / boundary points, the character arg-max of every point, and the complete outputs of four
queries.  The frame itself is NOT stored: tests rebuild it from gomatching_amd.synth.make_clip (seeded) and the harness
resize, exactly as this script does.
"""
import hashlib
import os
import sys
import time
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from gomatching_amd.config import setup_cfg                      # noqa: E402
from gomatching_amd.weights import synth_state_dict              # noqa: E402
from gomatching_amd.synth import make_clip                       # noqa: E402
from oracle import gom_oracle as O                               # noqa: E402
from oracle import ref_shim                                      # noqa: E402
from oracle.gen_golden import build_ref_deepsolo, _np, _maxdiff  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = {"c1": {"src_hw": (640, 640), "resize": False},         # BASELINE configs[0]: 640x640 fed directly
         "c2": {"src_hw": (720, 1280), "resize": True}}         # BASELINE configs[1]: 1280x720 -> 1000x1778
QUERIES = (0, 33, 66, 99)
SEED = 0                                                        # bench.py's weights
CLS_BIAS = -1.0                                                 # fixed (not calibrated): class logits spread around the threshold


def full_cfg():
    cfg = setup_cfg(builtin="icdar15")
    cfg.MODEL.DEVICE = "cpu"
    cfg.MODEL.TRANSFORMER.AUX_LOSS = False
    return cfg


def frame_for(case):
    """The case's network input: f32 [3, H, W], 0..255, in cfg.INPUT.FORMAT order -- what `GoMBatchPredictor.prepare` hands
    to the model for frame 0 of the seeded synthetic clip (C1: the 640x640 frame as it is)."""
    c = CASES[case]
    cfg = full_cfg()
    clip = make_clip(1, c["src_hw"][0], c["src_hw"][1], clip_id=0, num_rects=12)
    if not c["resize"]:
        return torch.as_tensor(clip[0].astype("float32").transpose(2, 0, 1)).contiguous()
    from gomatching_amd.predictor import GoMBatchPredictor
    inputs, _ = GoMBatchPredictor(cfg, None).prepare([clip[0][:, :, ::-1]])
    return inputs[0]["image"].contiguous()


def weights(cfg):
    return synth_state_dict(cfg, seed=SEED, cls_bias={"detection_transformer.ctrl_point_class.0.bias": CLS_BIAS})


def oracle_features(cfg, sd, image):
    mean = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1)
    std = torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1)
    with torch.no_grad():
        feats = O.resnet50(((image - mean) / std)[None], sd)
    return [feats[k] for k in ("res3", "res4", "res5")]


def summarise(out, nq, P):
    """The tolerance-comparable digest of the five outputs (each [1, nq, P, C])."""
    v = {k: out[k].reshape(nq, P, -1) for k in ("pred_logits", "pred_text_logits", "pred_ctrl_points", "pred_bd_points",
                                                "query_features")}
    q = list(QUERIES)
    return {"logit_mean": v["pred_logits"].mean(1).reshape(nq), "ctrl": v["pred_ctrl_points"], "bd": v["pred_bd_points"],
            "recs": v["pred_text_logits"].argmax(-1), "text_top2_gap": (lambda t: t[..., 0] - t[..., 1])(
                v["pred_text_logits"].topk(2, -1).values),
            "q_logits": v["pred_logits"][q], "q_text": v["pred_text_logits"][q], "q_feat": v["query_features"][q]}


def case_full(case):
    cfg = full_cfg()
    T = cfg.MODEL.TRANSFORMER
    sd = weights(cfg)
    image = frame_for(case)
    t0 = time.time()
    feats = oracle_features(cfg, sd, image)
    ref = build_ref_deepsolo(cfg, sd)
    pe = ref_shim.load("adet.layers.pos_encoding")
    misc = ref_shim.load("adet.utils.misc")
    masks = [torch.zeros(1, f.shape[2], f.shape[3], dtype=torch.bool) for f in feats]
    posenc = pe.PositionalEncoding2D(T.HIDDEN_DIM // 2, T.TEMPERATURE, normalize=True)
    nts = [misc.NestedTensor(f, m) for f, m in zip(feats, masks)]
    pos = [posenc(nt) for nt in nts]
    seen = []
    real_topk = torch.topk

    def spy(x, k, *a, **kw):                                    # the reference's own top-k call (deformable_transformer.py:188)
        r = real_topk(x, k, *a, **kw)
        seen.append((tuple(x.shape), k, r[1].clone(), r[0].clone(), real_topk(x, k + 1, *a, **kw)[0][..., -1].clone()))
        return r

    torch.topk = spy
    try:
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = ref(nts, list(pos), [None, posenc])
    finally:
        torch.topk = real_topk
    S = sum(f.shape[2] * f.shape[3] for f in feats) + ((feats[2].shape[2] + 1) // 2) * ((feats[2].shape[3] + 1) // 2)
    prop = [s for s in seen if s[1] == T.NUM_QUERIES and s[0][-1] == S]
    assert len(prop) == 1, [(s[0], s[1]) for s in seen]
    topk_idx, topk_val = prop[0][2].reshape(-1), prop[0][3].reshape(-1)
    margin = float(topk_val.min() - prop[0][4].reshape(-1)[0])  # last winner minus first loser
    print("full/%s reference ran in %.1f s: S = %d, top-k logit range %.4f .. %.4f, smallest gap inside the top-k %.3e, "
          "last winner - first loser %.3e" % (case, time.time() - t0, S, float(topk_val.max()), float(topk_val.min()),
                                              float((topk_val[:-1] - topk_val[1:]).min()), margin))
    out = {"hw": np.asarray(image.shape[-2:]), "S": np.asarray([S]), "seed": np.asarray([SEED]),
           "cls_bias": np.asarray([CLS_BIAS], np.float32), "queries": np.asarray(QUERIES),
           "topk_idx": _np(topk_idx).astype(np.int64), "topk_val": _np(topk_val), "topk_margin": np.asarray([margin], np.float32),
           "image_sha1": np.frombuffer(hashlib.sha1(_np(image).tobytes()).digest(), np.uint8)}
    for k in ("pred_logits", "pred_text_logits", "pred_ctrl_points", "pred_bd_points", "query_features"):
        out["sha1_" + k] = np.frombuffer(hashlib.sha1(np.ascontiguousarray(_np(r[k])).tobytes()).digest(), np.uint8)
    for k, v in summarise(r, T.NUM_QUERIES, T.NUM_POINTS).items():
        out[k] = _np(v)
    # oracle self-check on the same features
    with torch.no_grad():
        taps = {}
        mine = O.deepsolo_forward(sd, cfg, feats, masks, [O.pos_encoding_2d(m, T.HIDDEN_DIM // 2, T.TEMPERATURE) for m in masks],
                                  taps=taps)
    print("full/%s oracle-vs-reference: top-k identical %s; max|d| %s" % (
        case, bool(torch.equal(taps["topk"].reshape(-1), topk_idx)),
        {k: "%.2e" % _maxdiff(r[k], mine[k]) for k in r if r[k] is not None}))
    np.savez_compressed(os.path.join(GOLD, "full_%s.npz" % case), **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    for c in (sys.argv[1:] or list(CASES)):
        case_full(c)
