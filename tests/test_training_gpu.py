"""GPU: the device training path of the association head (gomatching_amd/training.py + csrc/train.hip) against the
reference's own losses and gradients (tests/golden/train_asso_{lst,pp}.npz, train_res_ic15.npz -- produced by the
reference's `_forward_asso` / `loss_res` in training mode, oracle/gen_golden_train.py)."""
import numpy as np
import pytest
import torch

from helpers import mini_cfg, golden
from gomatching_amd.weights import synth_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda"
GRAD_KEYS = {"lst": ["asso_head.fc2.weight", "long_term_matcher.decoder.layers.0.multihead_attn.in_proj_weight",
                     "short_term_matcher.encoder.layers.0.linear1.weight",
                     "long_term_matcher.encoder.layers.0.self_attn.out_proj.bias"],
             "pp": ["asso_head.fc2.weight", "shared_matcher.decoder.layers.0.multihead_attn.in_proj_weight",
                    "asso_head.fc1.bias", "shared_matcher.decoder.layers.0.multihead_attn.out_proj.weight"]}


def _params(cfg):
    sd = synth_state_dict(cfg, seed=7)
    return {k: torch.nn.Parameter(torch.as_tensor(v).float().to(DEV)) for k, v in sd.items() if k.startswith("roi_heads.")}


@pytest.mark.parametrize("tag,builtin", [("lst", "icdar15"), ("pp", "pp_dstext")])
def test_association_losses_and_gradients_match_the_reference(tag, builtin):
    from gomatching_amd import training
    g = golden("train_asso_%s.npz" % tag)
    cfg = mini_cfg(builtin)
    cfg.MODEL.ASSO_HEAD.DROPOUT = 0.0
    for ci in range(2):
        params = _params(cfg)
        frames, targets = [], []
        f = 0
        while "c%d_f%d_pb" % (ci, f) in g:
            p = lambda k: g["c%d_f%d_%s" % (ci, f, k)]
            frames.append({"image_size": (96, 128), "proposal_boxes": torch.as_tensor(p("pb")).to(DEV),
                           "objectness_logits": torch.as_tensor(p("obj")).to(DEV),
                           "query_features": torch.as_tensor(p("qf").astype(np.float32)).to(DEV)})
            targets.append({"image_size": (96, 128), "gt_boxes": torch.as_tensor(p("gt")), "gt_instance_ids": torch.as_tensor(p("ids"))})
            f += 1
        losses = training.asso_losses(params, cfg, frames, targets)
        for k in ("loss_long_asso", "loss_short_asso"):
            ref = float(g["c%d_%s" % (ci, k)])
            assert abs(float(losses[k]) - ref) <= 1e-4 * max(1.0, abs(ref)), (tag, ci, k, float(losses[k]), ref)
        (losses["loss_long_asso"] + losses["loss_short_asso"]).backward()
        torch.cuda.synchronize()
        for gk in GRAD_KEYS[tag]:
            grad = params["roi_heads." + gk].grad
            assert grad is not None, gk
            flat = grad.reshape(-1).cpu()
            sample = flat[::max(1, flat.numel() // 4096)][:4096].numpy()
            ref = g["c%d_gsample_%s" % (ci, gk)]
            scale = max(1.0, float(np.abs(ref).max()))
            assert float(np.abs(sample - ref).max()) <= 1e-4 * scale, (tag, ci, gk, float(np.abs(sample - ref).max()))
            gabs = float(g["c%d_gabs_%s" % (ci, gk)])
            assert abs(float(flat.double().abs().sum()) - gabs) <= 1e-3 * max(1.0, gabs), (tag, ci, gk)


def test_loss_res_and_gradient_match_the_reference():
    from gomatching_amd import training
    g = golden("train_res_ic15.npz")
    cfg = mini_cfg("icdar15")
    params = _params(cfg)
    qf = torch.as_tensor(g["qf"].astype(np.float32)).to(DEV)
    pts = torch.as_tensor(g["pts"]).to(DEV)
    targets = [{"labels": np.zeros((g["t%d_ctrl" % b].shape[0],), np.int64), "ctrl_points": g["t%d_ctrl" % b]} for b in range(2)]
    out = training.loss_res(params, cfg, qf, pts, targets)
    ref = float(g["loss_res"])
    assert abs(float(out["loss_res"]) - ref) <= 1e-4 * max(1.0, abs(ref)), (float(out["loss_res"]), ref)
    out["loss_res"].backward()
    torch.cuda.synchronize()
    gw = params["roi_heads.rescoring_head.weight"].grad.cpu().numpy()
    gb = params["roi_heads.rescoring_head.bias"].grad.cpu().numpy()
    assert float(np.abs(gw - g["grad_w"]).max()) <= 1e-4 * max(1.0, float(np.abs(g["grad_w"]).max()))
    assert float(np.abs(gb - g["grad_b"]).max()) <= 1e-4 * max(1.0, float(np.abs(g["grad_b"]).max()))


def test_wrapper_forward_returns_the_loss_dict_and_trains_only_the_head():
    """`model(batched_inputs)` of the META_ARCH wrapper in train() mode: finite losses with the reference's keys, gradients on
    roi_heads only, and an SGD step on them changes the next loss (the HIP model is rebuilt from the updated parameters)."""
    from gomatching_amd.compat.d2_register import GoMatchingMI355X
    from gomatching_amd.synth import make_clip
    from gomatching_amd.weights import expand_for_reference
    cfg = mini_cfg("icdar15", device="cuda")
    cfg.MODEL.ASSO_HEAD.DROPOUT = 0.0
    sd = synth_state_dict(cfg, seed=7, cls_bias={"detection_transformer.ctrl_point_class.0.bias": 0.8,
                                                  "roi_heads.rescoring_head.bias": 0.8})
    model = GoMatchingMI355X(cfg).to(DEV)
    model.load_state_dict(expand_for_reference(sd))
    model.train()
    hw = (96, 128)
    clip = make_clip(4, hw[0], hw[1], clip_id=2)
    rng = np.random.default_rng(0)
    batch = []
    for t, fr in enumerate(clip):
        boxes = np.array([[10 + 3 * t, 12, 40 + 3 * t, 30], [60, 40 + 2 * t, 100, 62 + 2 * t]], np.float32)
        ctrl = np.stack([np.stack([np.linspace(b[0], b[2], 25), np.full(25, (b[1] + b[3]) / 2)], -1) for b in boxes]).astype(np.float32)
        batch.append({"image": torch.as_tensor(fr.astype("float32").transpose(2, 0, 1)),
                      "instances": {"gt_boxes": torch.as_tensor(boxes), "gt_instance_ids": torch.tensor([1, 2]),
                                    "ctrl_points": torch.as_tensor(ctrl)}})
    losses = model(batch)
    assert set(losses) == {"loss_long_asso", "loss_short_asso", "loss_res"}
    total = sum(losses.values())
    assert torch.isfinite(total)
    total.backward()
    got = {n for n, p in model.named_parameters() if p.grad is not None and float(p.grad.abs().sum()) > 0}
    assert got and all(n.startswith("roi_heads.") for n in got)
    assert "roi_heads.rescoring_head.weight" in got
    with torch.no_grad():
        for p in model.parameters():
            if p.grad is not None:
                p -= 0.05 * p.grad
    again = sum(model(batch).values())
    assert torch.isfinite(again) and float(again) != float(total)
