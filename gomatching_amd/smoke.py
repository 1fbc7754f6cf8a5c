"""Tiny end-to-end invocation for __graft_entry__.smoke(): the 8-frame mini clip through the HIP path,
checked against the CPU oracle (test infrastructure; imported only here and in tests/bench)."""
import os
import sys

import numpy as np
import torch


def run_smoke():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
    from helpers import mini_cfg, golden, e2e_state_dict
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.synth import make_clip
    from oracle import gom_oracle as O

    g = golden("e2e_lst.npz")
    cfg = mini_cfg("icdar15", device="cuda:0")
    sd = e2e_state_dict(cfg, g)
    hw = tuple(int(v) for v in g["hw"])
    clip = make_clip(4, hw[0], hw[1], clip_id=1)
    images = [torch.as_tensor(f.astype("float32").transpose(2, 0, 1)) for f in clip]
    tc = {k: 0.0 for k in ("pre_process", "backbone", "detector", "rescore", "tracker", "short_match", "long_match")}
    model = GoMatching(cfg, sd, device="cuda:0")
    insts, id_count = model.batch_inference([{"image": im} for im in images], 0, 0, [], tc)
    cpu_cfg = mini_cfg("icdar15")
    with torch.no_grad():
        per_frame = []
        for im in images:
            per_frame.extend(O.detect_frames(sd, cpu_cfg, [im]))
        ref, ref_count = O.track_clip(sd, cpu_cfg, per_frame)
    assert int(id_count) == int(ref_count), (id_count, ref_count)
    worst = 0.0
    for a, b in zip(insts, ref):
        assert a.track_ids.cpu().tolist() == b["track_ids"].tolist()
        assert a.recs.cpu().tolist() == b["recs"].tolist()
        if len(a):
            worst = max(worst, float((a.bd.cpu() - b["bd"]).abs().max()))
    assert worst < 1e-3, worst
    print("smoke: 4-frame clip, dets/frame %s, ids/recs identical to the oracle, max|d bd| = %.2e px" % (
        [len(x) for x in insts], worst))
