"""CPU: the oracle (oracle/gom_oracle.py) against the fixtures produced by the reference's own
modules (oracle/gen_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from gomatching_amd.synth import make_clip
from gomatching_amd.weights import synth_state_dict
from oracle import gom_oracle as O
from helpers import mini_cfg, golden, e2e_state_dict, t

TOL = 2e-5


@pytest.mark.parametrize("case", ["enc", "dec", "oob"])
def test_msda_op(case):
    g = golden("msda.npz")
    out = O.ms_deform_attn_forward(t(g[case + "_value"]), t(g[case + "_shapes"]), t(g[case + "_lsi"]),
                                   t(g[case + "_loc"]), t(g[case + "_w"]))
    np.testing.assert_allclose(out.numpy(), g[case + "_out"], atol=TOL, rtol=0)


@pytest.mark.parametrize("builtin,tag,voc", [("icdar15", "ic15", None), ("bovtext", "voc96", 96)])
def test_deepsolo_mini(builtin, tag, voc):
    g = golden("deepsolo_%s.npz" % tag)
    cfg = mini_cfg(builtin, voc=voc)
    sd = synth_state_dict(cfg, seed=7)
    feats = [t(g["feat%d" % i]) for i in range(3)]
    masks = [torch.zeros(f.shape[0], f.shape[2], f.shape[3], dtype=torch.bool) for f in feats]
    T = cfg.MODEL.TRANSFORMER
    pos = [O.pos_encoding_2d(m, T.HIDDEN_DIM // 2, T.TEMPERATURE) for m in masks]
    for i in range(3):
        np.testing.assert_allclose(pos[i].numpy(), g["pos%d" % i], atol=1e-6, rtol=0)
    with torch.no_grad():
        out = O.deepsolo_forward(sd, cfg, feats, masks, pos)
    for k in ("pred_logits", "pred_text_logits", "pred_ctrl_points", "pred_bd_points", "query_features"):
        np.testing.assert_allclose(out[k].numpy(), g[k], atol=TOL, rtol=0, err_msg=k)


def test_deepsolo_padded_batch():
    """A batch padded to 64x96 around 41x70 images: padding masks reach the position tables, the encoder value
    rows, the valid ratios and the proposal validity (gom_lstmatcher.py:63-76)."""
    g = golden("deepsolo_padded.npz")
    cfg = mini_cfg("icdar15")
    sd = synth_state_dict(cfg, seed=7)
    feats = [t(g["feat%d" % i]) for i in range(3)]
    masks = [t(g["mask%d" % i]).bool() for i in range(3)]
    assert bool(masks[0].any()) and bool(masks[1].any())
    T = cfg.MODEL.TRANSFORMER
    pos = [O.pos_encoding_2d(m, T.HIDDEN_DIM // 2, T.TEMPERATURE) for m in masks]
    for i in range(3):
        np.testing.assert_allclose(pos[i].numpy(), g["pos%d" % i], atol=1e-6, rtol=0)
    with torch.no_grad():
        out = O.deepsolo_forward(sd, cfg, feats, masks, pos)
    for k in ("pred_logits", "pred_text_logits", "pred_ctrl_points", "pred_bd_points", "query_features"):
        np.testing.assert_allclose(out[k].numpy(), g[k], atol=TOL, rtol=0, err_msg=k)


def test_valid_shapes_follow_the_masks():
    """Host logic: DeepSolo.valid_shapes == the valid extents of the reference's masks, incl. the resampled 4th level."""
    from gomatching_amd.modeling.deepsolo import DeepSolo
    import torch.nn.functional as F
    for hw, pad in (((41, 70), (64, 96)), ((720, 1280), (736, 1280)), ((1000, 1777), (1024, 1792)), ((33, 33), (64, 64))):
        shapes, masks = [], []
        for s in (8, 16, 32):
            m = torch.ones(1, pad[0] // s, pad[1] // s, dtype=torch.bool)
            m[:, :-(-hw[0] // s), :-(-hw[1] // s)] = False
            masks.append(m)
            shapes.append(tuple(m.shape[1:]))
        h3, w3 = (shapes[2][0] - 1) // 2 + 1, (shapes[2][1] - 1) // 2 + 1
        shapes.append((h3, w3))
        masks.append(F.interpolate(masks[0][None].float(), size=(h3, w3)).to(torch.bool)[0])
        got = DeepSolo.valid_shapes(shapes, hw)
        want = [(int((~m[0, :, 0]).sum()), int((~m[0, 0, :]).sum())) for m in masks]
        assert [tuple(v) for v in got] == want, (hw, got, want)


@pytest.mark.parametrize("builtin,tag", [("icdar15", "lst"), ("pp_dstext", "pp")])
def test_matcher_heads(builtin, tag):
    g = golden("matcher_%s.npz" % tag)
    cfg = mini_cfg(builtin)
    sd = synth_state_dict(cfg, seed=7)
    with torch.no_grad():
        for n in (1, 7, 20):
            x = t(g["fc_in_%d" % n]).flatten(1)
            for k in range(2):
                x = torch.relu(O.linear(x, sd, "roi_heads.asso_head.fc%d" % (k + 1)))
            np.testing.assert_allclose(x.numpy(), g["fc_out_%d" % n], atol=TOL, rtol=0)
        for ci in range(5):
            n_t = [int(v) for v in g["asso%d_nt" % ci]]
            k, short = int(g["asso%d_k" % ci][0]), bool(g["asso%d_k" % ci][1])
            out = O.asso_scores(sd, cfg, t(g["asso%d_reid" % ci]), n_t, k, short)
            np.testing.assert_allclose(out.numpy(), g["asso%d_out" % ci], atol=TOL, rtol=0)


def _tracker_inputs(g):
    size = tuple(int(v) for v in g["image_size"])
    frames = int(g["num_frames"][0])
    return [O.Inst(size, reid_features=t(g["reid_%d" % f]).clone(), pred_boxes=t(g["boxes_%d" % f]).clone())
            for f in range(frames)], frames


@pytest.mark.parametrize("builtin,tag", [("icdar15", "lst"), ("pp_dstext", "pp")])
def test_tracker_trace(builtin, tag):
    """16-frame trace incl. an empty frame, births, drop-outs, long-term re-association, id_count quirk."""
    g = golden("tracker_%s.npz" % tag)
    cfg = mini_cfg(builtin)
    sd = synth_state_dict(cfg, seed=7)
    insts, frames = _tracker_inputs(g)
    with torch.no_grad():
        res, id_count = O.track_clip(sd, cfg, insts)
    assert int(id_count) == int(g["id_count"][0])
    for f in range(frames):
        assert res[f]["track_ids"].tolist() == g["ids_%d" % f].tolist(), f
    kept = O.remove_short_track(cfg, res)
    for f in range(frames):
        assert kept[f]["track_ids"].tolist() == g["kept_ids_%d" % f].tolist(), f


@pytest.mark.parametrize("builtin,tag", [("icdar15", "lst"), ("pp_dstext", "pp")])
def test_match_log_leaves_the_trace_alone_and_the_tie_rule_is_strict(builtin, tag):
    """oracle.MatchLog (margins of the tracker's discrete decisions) must not change a single id of the reference's trace, and
    tests/helpers.track_clip_tie_aware -- the rule the GPU clip tests compare ids under -- must (i) accept identical ids with
    nothing forced, (ii) accept ids that part from the oracle's at ONE decision only when that decision's gap is below eps, then
    requiring every later id to follow, (iii) reject the same ids at an eps below the gap."""
    from helpers import track_clip_tie_aware
    g = golden("tracker_%s.npz" % tag)
    cfg = mini_cfg(builtin)
    sd = synth_state_dict(cfg, seed=7)
    insts, frames = _tracker_inputs(g)
    want = [g["ids_%d" % f].tolist() for f in range(frames)]
    res, id_count, rep = track_clip_tie_aware(sd, cfg, insts, want)
    assert int(id_count) == int(g["id_count"][0]) and rep["forced"] == [] and rep["replays"] == 1
    assert rep["margins"]["matches"] >= frames - 2 and rep["margins"]["min_thr_margin"] > 0
    # the widest-margin-first alternative of some match as "another implementation's" ids: gap known, far above 1e-4
    import copy
    probe = O.MatchLog(eps=10.0)
    with torch.no_grad():
        O.track_clip(sd, cfg, copy.deepcopy(insts), log=probe)
    other_ids, gap = None, None
    for idx, c in enumerate(probe.calls):                               # (a forced short-term alternative is often undone by the
        for k in range(1, c["n_alt"] + 1):                              #  long-term match of the same frame: take one that shows)
            forced = O.MatchLog(eps=10.0, script={idx: k})
            with torch.no_grad():
                other, _ = O.track_clip(sd, cfg, copy.deepcopy(insts), log=forced)
            ids = [x["track_ids"].tolist() for x in other]
            if ids != want and forced.calls[idx]["picked_gap"] > 1e-3:
                other_ids, gap = ids, forced.calls[idx]["picked_gap"]
                break
        if other_ids is not None:
            break
    assert other_ids is not None
    _, _, rep = track_clip_tie_aware(sd, cfg, insts, other_ids, eps=gap * 1.01, max_runs=64)
    assert len(rep["forced"]) >= 1 and all(c["picked_gap"] < gap * 1.01 for c in rep["forced"])
    with pytest.raises(AssertionError):
        track_clip_tie_aware(sd, cfg, insts, other_ids, eps=min(gap * 0.5, 1e-4))


@pytest.mark.parametrize("builtin,tag", [("icdar15", "lst"), ("pp_dstext", "pp")])
def test_end_to_end_mini_clip(builtin, tag):
    """Whole path on 8 tiny frames: Bezier/boundary points within 1e-3 px-scaled, identical recs and ids."""
    g = golden("e2e_%s.npz" % tag)
    cfg = mini_cfg(builtin)
    sd = e2e_state_dict(cfg, g)
    hw = tuple(int(v) for v in g["hw"])
    frames = int(g["num_frames"][0])
    clip = make_clip(frames, hw[0], hw[1], clip_id=1)
    images = [torch.as_tensor(f.astype("float32").transpose(2, 0, 1)) for f in clip]
    res, id_count = O.run_clip(sd, cfg, images)
    assert int(id_count) == int(g["id_count"][0])
    for f in range(frames):
        r = res[f]["instances"]
        assert r["track_ids"].tolist() == g["track_ids_%d" % f].tolist()
        assert r["recs"].tolist() == g["recs_%d" % f].tolist()
        for k in ("scores", "bd", "ctrl_points", "pred_boxes"):
            np.testing.assert_allclose(r[k].numpy(), g["%s_%d" % (k, f)], atol=1e-3, rtol=0, err_msg=k)


@pytest.mark.parametrize("name", ["c1", "c2"])
def test_oracle_vs_reference_full_size(name):
    """the oracle against the reference's OWN DeepSolo module at full size -- C1 640x640 (S = 8 500) and C2
    1000x1778 (S = 37 171), 100 queries -- where the proposal stage's top-k runs over tens of thousands of near-tied class
    logits (deformable_transformer.py:183-199).  Fixture: oracle/gen_golden_full.py."""
    from full_fixture import case, compare
    g, cfg, sd, image = case(name)
    T = cfg.MODEL.TRANSFORMER
    mean = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1)
    std = torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1)
    with torch.no_grad():
        feats = O.resnet50(((image - mean) / std)[None], sd)
        feats = [feats[k] for k in ("res3", "res4", "res5")]
        masks = [torch.zeros(1, f.shape[2], f.shape[3], dtype=torch.bool) for f in feats]
        taps = {}
        out = O.deepsolo_forward(sd, cfg, feats, masks, [O.pos_encoding_2d(m, T.HIDDEN_DIM // 2, T.TEMPERATURE) for m in masks],
                                 taps=taps)
    assert int(g["S"][0]) == taps["memory"].shape[1]
    err, moved = compare(g, out, taps["topk"].reshape(-1), T.NUM_QUERIES, T.NUM_POINTS, tol=2e-5, what="oracle " + name)
    assert moved == 0


@pytest.mark.parametrize("case", ["odd_f64", "odd_f32", "wide_f64", "one_f32", "ship_f64", "ship_f32"])
def test_msda_general_form_and_gradients(case):
    """Any heads / channels / levels / points, fp32 and fp64: the oracle's op and its autograd gradients against the outputs
    and gradients of the reference's own ms_deform_attn_core_pytorch (oracle/gen_golden_msda_any.py)."""
    g = golden("msda_any.npz")
    v, loc, w = (t(g[case + k]).requires_grad_(True) for k in ("_value", "_loc", "_w"))
    out = O.ms_deform_attn_forward(v, t(g[case + "_shapes"]), t(g[case + "_lsi"]), loc, w)
    gv, gl, gw = torch.autograd.grad(out, (v, loc, w), t(g[case + "_gout"]))
    tol = 1e-12 if v.dtype == torch.float64 else TOL
    for got, key in ((out.detach(), "_out"), (gv, "_grad_value"), (gl, "_grad_loc"), (gw, "_grad_w")):
        exp = g[case + key]
        assert got.numpy().dtype == exp.dtype
        np.testing.assert_allclose(got.numpy(), exp, atol=tol * max(1.0, float(np.abs(exp).max())), rtol=0)
