"""Shared builders for the parity tests (configs/weights identical to oracle/gen_golden.py)."""
import os

import numpy as np
import torch

from gomatching_amd.config import setup_cfg
from gomatching_amd.weights import synth_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
MINI_NQ = 12


def mini_cfg(builtin="icdar15", nq=MINI_NQ, voc=None, device="cpu"):
    cfg = setup_cfg(builtin=builtin)
    cfg.MODEL.DEVICE = device
    cfg.MODEL.TRANSFORMER.NUM_QUERIES = nq
    if voc is not None:
        cfg.MODEL.TRANSFORMER.VOC_SIZE = voc
    return cfg


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def e2e_state_dict(cfg, gold):
    bias = {"detection_transformer.ctrl_point_class.0.bias": float(gold["cls_bias"][0])}
    if cfg.MODEL.ROI_HEADS.WITH_RESR:
        bias["roi_heads.rescoring_head.bias"] = float(gold["cls_bias"][1])
    return synth_state_dict(cfg, seed=7, cls_bias=bias)


def t(x):
    return torch.from_numpy(np.asarray(x))


TIE_EPS = 1e-4


def track_clip_tie_aware(sd, cfg, raw, want_ids, eps=TIE_EPS, max_runs=24):
    """The oracle's tracker over the detections `raw` (a list of oracle `Inst`s: copied for every replay), compared with
    `want_ids` (per frame, the ids another implementation gave the SAME detections).  north_star's bar is identical track
    assignment; the tracker's decisions are discrete (linear-sum assignment on -traj, then `traj > thr`:
    gom_lstmatcher.py:434-452, 521-554), so two correct fp32 evaluations can part at a near-tie.  Rule (the tracker's twin of
    test_clips_fullsize_gpu._rank_swaps): the ids must be identical, OR every decision at which they part sits on a gap below `eps`
    in the ORACLE's own traj matrix (assignment total vs the best different assignment, or |traj - thr|), and the oracle replayed
    with exactly those decisions taken the other way reproduces EVERY id of every later frame.
    -> (instances, id_count, report); report["forced"] lists the decisions taken the other way with their gaps (empty = plain
    identity), report["margins"] the smallest assignment gap / threshold margin over the clip."""
    import copy
    from oracle import gom_oracle as O
    want = [list(map(int, w)) for w in want_ids]
    runs = [0]

    def replay(script):
        runs[0] += 1
        log = O.MatchLog(eps=eps, script=script)
        with torch.no_grad():
            inst, count = O.track_clip(sd, cfg, copy.deepcopy(raw), log=log)
        got = [x["track_ids"].tolist() for x in inst]
        bad = [f for f, (g, w) in enumerate(zip(got, want)) if g != w]
        return inst, count, log, (bad[0] if bad else None)

    def search(script):
        inst, count, log, f = replay(script)
        if f is None:
            return inst, count, log
        for idx, c in enumerate(log.calls):                             # a decision of the first frame that differs, not yet forced
            if c["frame"] != f or idx in script:
                continue
            for k in range(1, c["n_alt"] + 1):
                if runs[0] >= max_runs:
                    return None
                res = search({**script, idx: k})
                if res is not None:
                    return res
        return None

    first = replay({})
    res = (first[0], first[1], first[2]) if first[3] is None else search({})
    assert res is not None, ("track ids differ from the oracle's at frame %d and no decision of that frame sits on a gap below %g "
                             "(margins of that frame: %s)" % (first[3], eps, [c for c in first[2].calls if c["frame"] == first[3]]))
    inst, count, log = res
    forced = [dict(c, call=i) for i, c in enumerate(log.calls) if c["picked"]]
    assert all(c["picked_gap"] < eps for c in forced)
    return inst, count, {"forced": forced, "margins": first[2].summary(), "replays": runs[0]}
