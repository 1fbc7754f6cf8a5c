"""Training forward of the association head: the CPU restatement (oracle/train_oracle.py) against the
losses and gradients the reference's own `_forward_asso` produced in training mode (oracle/gen_golden_train.py)."""
import numpy as np
import pytest
import torch

from helpers import golden, mini_cfg
from gomatching_amd.weights import synth_state_dict
from oracle import train_oracle

GRAD_KEYS = {"lst": ["asso_head.fc2.weight", "long_term_matcher.decoder.layers.0.multihead_attn.in_proj_weight",
                     "short_term_matcher.encoder.layers.0.linear1.weight",
                     "long_term_matcher.encoder.layers.0.self_attn.out_proj.bias"],
             "pp": ["asso_head.fc2.weight", "shared_matcher.decoder.layers.0.multihead_attn.in_proj_weight", "asso_head.fc1.bias",
                    "shared_matcher.decoder.layers.0.multihead_attn.out_proj.weight"]}


def _clip(g, ci, size=(96, 128)):
    props, targets, f = [], [], 0
    while "c%d_f%d_pb" % (ci, f) in g:
        k = "c%d_f%d_" % (ci, f)
        props.append({"image_size": size, "proposal_boxes": torch.from_numpy(g[k + "pb"]).float(),
                      "objectness_logits": torch.from_numpy(g[k + "obj"]).float(),
                      "query_features": torch.from_numpy(g[k + "qf"].astype(np.float32))})
        targets.append({"image_size": size, "gt_boxes": torch.from_numpy(g[k + "gt"]).float(),
                        "gt_instance_ids": torch.from_numpy(g[k + "ids"]).long()})
        f += 1
    return props, targets


@pytest.mark.parametrize("builtin,tag", [("icdar15", "lst"), ("pp_dstext", "pp")])
@pytest.mark.parametrize("ci", [0, 1])
def test_association_losses_and_gradients(builtin, tag, ci):
    g = golden("train_asso_%s.npz" % tag)
    cfg = mini_cfg(builtin)
    sd = synth_state_dict(cfg, seed=7)
    props, targets = _clip(g, ci)
    assert len(props) == 5 and any(len(p["proposal_boxes"]) == 0 for p in props) == (ci == 0)     # case 0 holds the empty frame
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("roi_heads.")}
    losses = train_oracle.asso_losses({**sd, **params}, cfg, props, targets)
    for k in ("loss_long_asso", "loss_short_asso"):
        assert abs(float(losses[k]) - float(g["c%d_%s" % (ci, k)])) <= 2e-5 * max(1.0, abs(float(g["c%d_%s" % (ci, k)]))), k
    (losses["loss_long_asso"] + losses["loss_short_asso"]).backward()
    for gk in GRAD_KEYS[tag]:
        grad = params["roi_heads." + gk].grad.reshape(-1)
        sample = grad[::max(1, grad.numel() // 4096)][:4096].numpy()
        ref = g["c%d_gsample_%s" % (ci, gk)]
        assert np.abs(sample - ref).max() <= 2e-6 * max(1.0, float(np.abs(ref).max())) + 1e-6, gk
        assert abs(float(grad.double().abs().sum()) - float(g["c%d_gabs_%s" % (ci, gk)])) <= 1e-4 * float(g["c%d_gabs_%s" % (ci, gk)]), gk


def test_no_ground_truth_means_zero_losses():
    cfg = mini_cfg("icdar15")
    sd = synth_state_dict(cfg, seed=7)
    g = golden("train_asso_lst.npz")
    props, targets = _clip(g, 1)
    for t in targets:
        t["gt_boxes"], t["gt_instance_ids"] = torch.zeros(0, 4), torch.zeros(0, dtype=torch.long)
    out = train_oracle.asso_losses(sd, cfg, props, targets)
    assert float(out["loss_long_asso"]) == 0.0 and float(out["loss_short_asso"]) == 0.0


def test_association_ground_truth_table():
    """_get_asso_gt on a hand-made case: proposal index per (track, frame), background = n_t[t], one cue per proposal."""
    pb = torch.tensor([[0.0, 0.0, 0.2, 0.2], [0.5, 0.5, 0.7, 0.7], [0.0, 0.0, 0.21, 0.2], [0.8, 0.8, 0.9, 0.9]])
    pt = torch.tensor([0, 0, 1, 1])
    tb = torch.tensor([[0.0, 0.0, 0.2, 0.2], [0.5, 0.5, 0.7, 0.7], [0.0, 0.0, 0.2, 0.2]])
    tt = torch.tensor([0, 0, 1])
    ids = torch.tensor([4, 9, 4])
    gt, cues = train_oracle.asso_gt(pb, pt, tb, tt, ids, [2, 2])
    assert gt.tolist() == [[0, 0], [1, 2]]                       # track 9 is not seen in frame 1 -> background (= n_t = 2)
    assert cues.tolist() == [0, 1, 0, -1]


def test_rescoring_loss_and_gradient():
    """loss_res: Hungarian matching of control points (CtrlPointHungarianMatcher4GM) + sigmoid focal loss of the rescoring
    head, against the reference's own value and gradient."""
    g = golden("train_res_ic15.npz")
    cfg = mini_cfg("icdar15")
    sd = synth_state_dict(cfg, seed=7)
    qf, pts = torch.from_numpy(g["qf"].astype(np.float32)), torch.from_numpy(g["pts"])
    targets = []
    b = 0
    while "t%d_ctrl" % b in g:
        c = torch.from_numpy(g["t%d_ctrl" % b])
        targets.append({"labels": torch.zeros(c.shape[0], dtype=torch.long), "ctrl_points": c})
        b += 1
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("roi_heads.rescoring")}
    out = train_oracle.loss_res({**sd, **params}, cfg, qf, pts, targets)
    assert abs(float(out["loss_res"].detach()) - float(g["loss_res"])) <= 2e-6
    out["loss_res"].backward()
    assert float((params["roi_heads.rescoring_head.weight"].grad - torch.from_numpy(g["grad_w"])).abs().max()) <= 1e-6
    assert float((params["roi_heads.rescoring_head.bias"].grad - torch.from_numpy(g["grad_b"])).abs().max()) <= 1e-6
    idx = train_oracle.point_matching(cfg, O_linear(qf, sd), pts, targets)
    assert [len(i) for i, _ in idx] == [len(t["labels"]) for t in targets]


def O_linear(qf, sd):
    from oracle import gom_oracle as O
    return O.linear(qf, sd, "roi_heads.rescoring_head")


@pytest.mark.parametrize("tag", ["lst", "pp"])
def test_host_association_targets_equal_the_oracle(tag):
    """gomatching_amd.training.association_targets (numpy, product side) against the pinned torch restatement on the
    fixture clips (false positives, a missed object, an empty frame) and on random boxes."""
    from gomatching_amd import training
    g = golden("train_asso_%s.npz" % tag)
    cfg = mini_cfg("icdar15" if tag == "lst" else "pp_dstext")
    for ci in (0, 1):
        props, targets = _clip(g, ci)
        keep = [p["objectness_logits"] > cfg.MODEL.ASSO_HEAD.ASSO_THRESH for p in props]
        fr = [{"image_size": p["image_size"], "proposal_boxes": p["proposal_boxes"][k]} for p, k in zip(props, keep)]
        n_t = [len(f["proposal_boxes"]) for f in fr]
        pb, pt = train_oracle._boxes_time(fr, "proposal_boxes")
        tb, tt = train_oracle._boxes_time(targets, "gt_boxes")
        ids = torch.cat([t["gt_instance_ids"] for t in targets if len(t["gt_boxes"]) > 0])
        want_gt, want_cues = train_oracle.asso_gt(pb, pt, tb, tt, ids, n_t)
        sizes = [p["image_size"] for p in props]
        npb, npt = training.normalised_boxes_and_times([f["proposal_boxes"].numpy() for f in fr], sizes)
        ntb, ntt = training.normalised_boxes_and_times([t["gt_boxes"].numpy() for t in targets], sizes)
        assert np.array_equal(npb, pb.numpy()) and np.array_equal(npt, pt.numpy())
        got_gt, got_cues = training.association_targets(npb, npt, ntb, ntt, ids.numpy(), n_t)
        assert got_gt.tolist() == want_gt.tolist() and got_cues.tolist() == want_cues.tolist()
    rng = np.random.default_rng(3)
    a = rng.random((40, 2)).astype(np.float32)
    b = rng.random((30, 2)).astype(np.float32)
    A = np.concatenate([a, a + rng.random((40, 2)).astype(np.float32) * 0.3], 1)
    B = np.concatenate([b, b + rng.random((30, 2)).astype(np.float32) * 0.3], 1)
    ref = train_oracle._pairwise_iou(torch.from_numpy(A), torch.from_numpy(B)).numpy()
    assert np.abs(training.pairwise_iou(A, B) - ref).max() <= 1e-7


def test_host_point_matching_equals_the_oracle():
    """Product-side Hungarian matching of control points (numpy cost + the library's C++ LSA) against the pinned oracle."""
    from gomatching_amd import training
    g = golden("train_res_ic15.npz")
    cfg = mini_cfg("icdar15")
    sd = synth_state_dict(cfg, seed=7)
    qf, pts = torch.from_numpy(g["qf"].astype(np.float32)), torch.from_numpy(g["pts"])
    targets, b = [], 0
    while "t%d_ctrl" % b in g:
        c = torch.from_numpy(g["t%d_ctrl" % b])
        targets.append({"labels": torch.zeros(c.shape[0], dtype=torch.long), "ctrl_points": c})
        b += 1
    logits = O_linear(qf, sd)
    want = train_oracle.point_matching(cfg, logits, pts, targets)
    L = cfg.MODEL.TRANSFORMER.LOSS
    for i, t in enumerate(targets):
        r, c = training.point_matching(logits[i].numpy(), pts[i].numpy(), t["ctrl_points"].numpy(), L.FOCAL_ALPHA, L.FOCAL_GAMMA,
                                       L.POINT_CLASS_WEIGHT, L.POINT_COORD_WEIGHT)
        assert r.tolist() == want[i][0].tolist() and c.tolist() == want[i][1].tolist()
