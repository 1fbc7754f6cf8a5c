"""Generate tests/golden/*.npz by running the REFERENCE's own modules (container-only).

TEST INFRASTRUCTURE.  Run in the build container, where /root/reference exists:

    python -m oracle.gen_golden            # writes tests/golden/*.npz and prints oracle-vs-reference diffs

The reference's arithmetic files are imported unmodified through oracle/ref_shim.py, loaded with
the repo's synthetic weights (gomatching_amd/weights.py) and executed on CPU; only their inputs
and outputs (small arrays) are committed.  No reference file travels in any form.
"""
import copy
import os
import sys
import warnings

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from gomatching_amd.config import setup_cfg                      # noqa: E402
from gomatching_amd.weights import synth_state_dict, expand_for_reference  # noqa: E402
from gomatching_amd.synth import make_clip                       # noqa: E402
from oracle import gom_oracle as O                               # noqa: E402
from oracle import ref_shim                                      # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# cls biases calibrated once (see calibrate()) so that a useful fraction of queries passes the threshold
MINI_NQ = 12


def mini_cfg(builtin="icdar15", nq=MINI_NQ, voc=None):
    cfg = setup_cfg(builtin=builtin)
    cfg.MODEL.DEVICE = "cpu"
    cfg.MODEL.TRANSFORMER.NUM_QUERIES = nq
    if voc is not None:
        cfg.MODEL.TRANSFORMER.VOC_SIZE = voc
    cfg.MODEL.TRANSFORMER.AUX_LOSS = False
    return cfg


def _np(x):
    return x.detach().cpu().numpy()


def _maxdiff(a, b):
    return float((a.double() - b.double()).abs().max()) if a.numel() else 0.0


def build_ref_deepsolo(cfg, sd):
    mod = ref_shim.load("adet.modeling.model.detection_transformer_wobackbone")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = mod.DETECTION_TRANSFORMER_WOBACKBONE(cfg)
    full = expand_for_reference(sd, cfg.MODEL.TRANSFORMER.DEC_LAYERS)
    mine = {k[len("detection_transformer."):]: v for k, v in full.items() if k.startswith("detection_transformer.")}
    missing, unexpected = m.load_state_dict(mine, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return m.eval()


def build_ref_roi_heads(cfg, sd):
    name = cfg.MODEL.ROI_HEADS.NAME
    modname = {"LSTMatcher": "lstmatcher", "SHA_FFN_CRSATTN": "shared_ffn_crsattn"}[name]
    mod = ref_shim.load("gomatching.modeling.roi_heads." + modname)
    m = getattr(mod, name)(cfg, None)
    mine = {k[len("roi_heads."):]: v for k, v in sd.items() if k.startswith("roi_heads.")}
    missing, unexpected = m.load_state_dict(mine, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return m.eval()


def build_ref_gomatching(cfg, sd):
    """GoMatching with the reference's detection_transformer / roi_heads / PositionalEncoding2D and a
    stand-in D2 backbone (the oracle's R-50 restatement; Detectron2 is not available)."""
    gm = ref_shim.load("gomatching.modeling.meta_arch.gom_lstmatcher")
    pe = ref_shim.load("adet.layers.pos_encoding")
    misc = ref_shim.load("adet.utils.misc")

    class FakeMasked(nn.Module):
        feature_strides = [8, 16, 32]

        def forward(self, images):
            feats = O.resnet50(images.tensor, sd)
            masks = gm.MaskedBackbone.mask_out_padding(
                self, [f.shape for f in feats.values()], images.image_sizes, images.tensor.device)
            return {k: misc.NestedTensor(f, m) for (k, f), m in zip(feats.items(), masks)}

    model = gm.GoMatching.__new__(gm.GoMatching)
    nn.Module.__init__(model)
    V = cfg.VIDEO_TEST
    model.test_len = cfg.INPUT.VIDEO.TEST_LEN
    model.overlap_thresh = V.OVERLAP_THRESH
    model.min_track_len = V.MIN_TRACK_LEN
    model.max_center_dist = V.MAX_CENTER_DIST
    model.decay_time = V.DECAY_TIME
    model.asso_thresh = cfg.MODEL.ASSO_HEAD.ASSO_THRESH
    model.with_iou = V.WITH_IOU
    model.local_no_iou = V.LOCAL_NO_IOU
    model.local_iou_only = V.LOCAL_IOU_ONLY
    model.not_mult_thresh = V.NOT_MULT_THRESH
    model.nms_thresh = V.NMS_THRESH
    model.with_rescore = cfg.MODEL.ROI_HEADS.WITH_RESR
    model.cfg = cfg
    model.device = torch.device("cpu")
    model.test_score_threshold = cfg.MODEL.TRANSFORMER.INFERENCE_TH_TEST
    model.min_size_test = None
    model.max_size_test = None
    T = cfg.MODEL.TRANSFORMER
    model.backbone = gm.Joiner(FakeMasked(), pe.PositionalEncoding2D(T.HIDDEN_DIM // 2, T.TEMPERATURE, normalize=True))
    model.detection_transformer = build_ref_deepsolo(cfg, sd)
    model.roi_heads = build_ref_roi_heads(cfg, sd)
    mean = torch.Tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1)
    std = torch.Tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1)
    model.normalizer = lambda x: (x - mean) / std
    return model.eval()


def new_time_cost():
    return {k: 0.0 for k in ("pre_process", "backbone", "detector", "rescore", "tracker", "short_match",
                             "long_match", "post_process", "total_time")}


# ------------------------------------------------------------------ cases
def case_msda():
    """(i) the native op, via the reference's own pure-PyTorch core (ms_deform_attn.py:40-60)."""
    msda = ref_shim.load("adet.layers.ms_deform_attn")
    g = torch.Generator().manual_seed(1234)
    out = {}
    shapes_list = {
        "enc": ([(6, 9), (3, 5), (2, 3), (1, 2)], None),
        "dec": ([(6, 9), (3, 5), (2, 3), (1, 2)], 50),
        "oob": ([(5, 7), (3, 4), (2, 2), (1, 1)], 40),
    }
    for name, (shapes, lq) in shapes_list.items():
        S = sum(h * w for h, w in shapes)
        Lq = S if lq is None else lq
        B, M, D, L, P = 2, 8, 32, 4, 4
        value = torch.randn(B, S, M, D, generator=g)
        if name == "oob":
            loc = torch.rand(B, Lq, M, L, P, 2, generator=g) * 1.6 - 0.3     # well outside [0,1]
        else:
            loc = torch.rand(B, Lq, M, L, P, 2, generator=g)
        w = torch.softmax(torch.randn(B, Lq, M, L * P, generator=g), -1).view(B, Lq, M, L, P)
        ss = torch.as_tensor(shapes, dtype=torch.long)
        lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
        ref = msda.ms_deform_attn_core_pytorch(value, shapes, loc, w)
        mine = O.ms_deform_attn_forward(value, ss, lsi, loc, w)
        print("msda/%s  oracle-vs-reference max|d| = %.3e" % (name, _maxdiff(ref, mine)))
        out.update({name + "_value": _np(value), name + "_shapes": _np(ss), name + "_lsi": _np(lsi),
                    name + "_loc": _np(loc), name + "_w": _np(w), name + "_out": _np(ref)})
    np.savez_compressed(os.path.join(GOLD, "msda.npz"), **out)


def case_deepsolo(builtin, tag, voc=None, image_hw=None):
    """(ii) DeepSolo-without-backbone at a mini geometry, random features in place of R-50 outputs.  `image_hw`: the
    unpadded image size of a batch padded to 64x96 -> non-trivial padding masks (gom_lstmatcher.py:63-76)."""
    cfg = mini_cfg(builtin, voc=voc)
    sd = synth_state_dict(cfg, seed=7)
    ref = build_ref_deepsolo(cfg, sd)
    pe = ref_shim.load("adet.layers.pos_encoding")
    misc = ref_shim.load("adet.utils.misc")
    g = torch.Generator().manual_seed(99)
    B = 2
    dims = [(512, 8, 12), (1024, 4, 6), (2048, 2, 3)]
    feats = [torch.randn(B, c, h, w, generator=g) * 0.5 for c, h, w in dims]
    masks = [torch.zeros(B, h, w, dtype=torch.bool) for _, h, w in dims]
    if image_hw is not None:
        for m, stride in zip(masks, (8, 16, 32)):
            m[:] = True
            m[:, :int(np.ceil(image_hw[0] / stride)), :int(np.ceil(image_hw[1] / stride))] = False
    T = cfg.MODEL.TRANSFORMER
    posenc = pe.PositionalEncoding2D(T.HIDDEN_DIM // 2, T.TEMPERATURE, normalize=True)
    nts = [misc.NestedTensor(f, m) for f, m in zip(feats, masks)]
    pos = [posenc(nt) for nt in nts]
    with torch.no_grad():
        r = ref(nts, list(pos), [None, posenc])
        taps = {}
        mine = O.deepsolo_forward(sd, cfg, feats, masks, [O.pos_encoding_2d(m, T.HIDDEN_DIM // 2, T.TEMPERATURE)
                                                           for m in masks], taps=taps)
    for k in r:
        if r[k] is not None:
            print("deepsolo/%s %-18s oracle-vs-reference max|d| = %.3e" % (tag, k, _maxdiff(r[k], mine[k])))
    print("deepsolo/%s pos_encoding max|d| = %.3e" % (tag, max(
        _maxdiff(a, O.pos_encoding_2d(m, T.HIDDEN_DIM // 2, T.TEMPERATURE)) for a, m in zip(pos, masks))))
    out = {"feat%d" % i: _np(f) for i, f in enumerate(feats)}
    out.update({"pos%d" % i: _np(p) for i, p in enumerate(pos)})
    out.update({k: _np(v) for k, v in r.items() if v is not None})
    # intermediate taps come from the (just validated) oracle; final outputs above are the reference's
    for k in ("memory", "enc_class", "topk", "init_ref", "enc0", "dec0"):
        out["tap_" + k] = _np(taps[k])
    for i, m in enumerate(masks):
        out["mask%d" % i] = _np(m)
    if image_hw is not None:
        out["image_hw"] = np.asarray(image_hw)
    np.savez_compressed(os.path.join(GOLD, "deepsolo_%s.npz" % tag), **out)


def _fake_instances(gm_mod, n_list, F, g, image_size):
    """Instances with reid/boxes as produced by roi_heads (for the tracker-only cases)."""
    Instances = ref_shim.Instances
    Boxes = ref_shim.Boxes
    res = []
    for n in n_list:
        inst = Instances(image_size)
        xy = torch.rand(n, 2, generator=g) * torch.tensor([image_size[1] * 0.7, image_size[0] * 0.7])
        wh = torch.rand(n, 2, generator=g) * 30 + 8
        inst.reid_features = torch.relu(torch.randn(n, F, generator=g))
        inst.pred_boxes = Boxes(torch.cat([xy, xy + wh], 1))
        inst.scores = torch.rand(n, generator=g)
        inst.pred_classes = torch.zeros(n, dtype=torch.long)
        inst.ctrl_points = torch.rand(n, 50, generator=g)
        inst.recs = torch.randint(0, 37, (n, 25), generator=g)
        inst.bd = torch.rand(n, 25, 4, generator=g)
        res.append(inst)
    return res


def case_matcher(builtin, tag):
    """(iv) FCHead4Query + matcher transformer + _activate_asso for n in {0,1,7,...}."""
    cfg = mini_cfg(builtin)
    sd = synth_state_dict(cfg, seed=7)
    rh = build_ref_roi_heads(cfg, sd)
    g = torch.Generator().manual_seed(5)
    out = {}
    with torch.no_grad():
        for n in (1, 7, 20):
            q = torch.randn(n, 25, 256, generator=g)
            ref = rh.asso_head(q)
            x = q.flatten(1)
            for k in range(2):
                x = torch.relu(O.linear(x, sd, "roi_heads.asso_head.fc%d" % (k + 1)))
            print("matcher/%s fchead n=%d max|d| = %.3e" % (tag, n, _maxdiff(ref, x)))
            out["fc_in_%d" % n] = _np(q)
            out["fc_out_%d" % n] = _np(ref)
        for ci, (n_t, k, short) in enumerate([([5, 7], 1, True), ([3, 0, 4, 6, 2, 5], 5, False),
                                              ([1, 1], 1, True), ([0, 4], 1, True), ([4, 9, 6], 2, False)]):
            N = sum(n_t)
            reid = torch.relu(torch.randn(N, 1024, generator=g)) * 0.5
            insts = _fake_instances(None, n_t, 1024, g, (96, 128))
            asso, _, _, _ = rh._forward_transformer(insts, reid[None], k, short_term=short)
            ref = torch.cat(rh._activate_asso(asso[-1].split(n_t, dim=1)), dim=1)
            mine = O.asso_scores(sd, cfg, reid, n_t, k, short)
            print("matcher/%s asso case%d max|d| = %.3e" % (tag, ci, _maxdiff(ref, mine)))
            out["asso%d_reid" % ci] = _np(reid)
            out["asso%d_nt" % ci] = np.asarray(n_t)
            out["asso%d_k" % ci] = np.asarray([k, int(short)])
            out["asso%d_logits" % ci] = _np(asso[-1])
            out["asso%d_out" % ci] = _np(ref)
    np.savez_compressed(os.path.join(GOLD, "matcher_%s.npz" % tag), **out)


def case_tracker(builtin, tag, frames=16):
    """(v) the reference's unmodified batch_inference/run_*_match/_remove_short_track on synthetic
    detections (inference() substituted), recording every id decision."""
    cfg = mini_cfg(builtin)
    sd = synth_state_dict(cfg, seed=7)
    model = build_ref_gomatching(cfg, sd)
    g = torch.Generator().manual_seed(11)
    size = (96, 128)
    # persistent objects with drifting boxes + per-object embedding, some drop-outs / births
    nobj = 9
    base = torch.relu(torch.randn(nobj + 6, 1024, generator=g))
    xy0 = torch.rand(nobj + 6, 2, generator=g) * torch.tensor([90.0, 60.0])
    vel = (torch.rand(nobj + 6, 2, generator=g) - 0.5) * 4
    wh = torch.rand(nobj + 6, 2, generator=g) * 20 + 10
    per_frame = []
    Instances, Boxes = ref_shim.Instances, ref_shim.Boxes
    for t in range(frames):
        alive = [o for o in range(nobj) if not (o % 4 == 1 and t % 5 == 2)]
        if t >= 6:
            alive += [nobj + (t - 6) // 3] if (nobj + (t - 6) // 3) < nobj + 6 else []
        if t == 9:
            alive = []                                      # an empty frame
        idx = torch.tensor(alive, dtype=torch.long)
        perm = torch.randperm(len(alive), generator=g)
        idx = idx[perm] if len(alive) else idx
        inst = Instances(size)
        n = len(idx)
        xy = xy0[idx] + vel[idx] * t
        inst.reid_features = torch.relu(base[idx] + 0.15 * torch.randn(n, 1024, generator=g))
        inst.pred_boxes = Boxes(torch.cat([xy, xy + wh[idx]], 1).reshape(n, 4))
        inst.scores = torch.rand(n, generator=g)
        inst.pred_classes = torch.zeros(n, dtype=torch.long)
        inst.ctrl_points = torch.rand(n, 50, generator=g)
        inst.recs = torch.randint(0, 37, (n, 25), generator=g)
        inst.bd = torch.rand(n, 25, 4, generator=g)
        per_frame.append(inst)
    saved = copy.deepcopy(per_frame)
    it = iter(per_frame)
    model.inference = lambda batched_inputs, time_cost: [next(it)]
    with torch.no_grad():
        insts, id_count = model.batch_inference([{} for _ in range(frames)], 0, 0, [], new_time_cost())
        ids_before = [x.track_ids.clone() for x in insts]
        insts = model._remove_short_track(insts)
    out = {"num_frames": np.asarray([frames]), "id_count": np.asarray([int(id_count)]),
           "image_size": np.asarray(size)}
    for t in range(frames):
        out["reid_%d" % t] = _np(saved[t].reid_features)
        out["boxes_%d" % t] = _np(saved[t].pred_boxes.tensor)
        out["ids_%d" % t] = _np(ids_before[t])
        out["kept_ids_%d" % t] = _np(insts[t].track_ids)
    # oracle self-check
    mine = [O.Inst(size, reid_features=s.reid_features.clone(), pred_boxes=s.pred_boxes.tensor.clone(),
                   scores=s.scores, pred_classes=s.pred_classes, ctrl_points=s.ctrl_points, recs=s.recs,
                   bd=s.bd) for s in saved]
    with torch.no_grad():
        mi, mc = O.track_clip(sd, cfg, mine)
    ok = all(torch.equal(a["track_ids"], b) for a, b in zip(mi, ids_before)) and int(mc) == int(id_count)
    print("tracker/%s ids identical oracle-vs-reference: %s (id_count %d, ids/frame %s)" % (
        tag, ok, int(id_count), [len(x) for x in ids_before]))
    mk = O.remove_short_track(cfg, mi)
    ok2 = all(torch.equal(a["track_ids"], b.track_ids) for a, b in zip(mk, insts))
    print("tracker/%s short-track removal identical: %s" % (tag, ok2))
    np.savez_compressed(os.path.join(GOLD, "tracker_%s.npz" % tag), **out)


def calibrate(cfg, sd, images, frac=0.35):
    """Pick ctrl_point_class / rescoring biases so that ~frac of the queries pass the threshold."""
    taps = {}
    with torch.no_grad():
        O.detect_frames(sd, cfg, images[:1], taps=taps)
    thr = cfg.MODEL.TRANSFORMER.INFERENCE_TH_TEST
    logit_thr = float(np.log(thr / (1 - thr)))
    m = taps["out_pred_logits"].mean(-2).flatten()
    shift = logit_thr - float(torch.quantile(m, 1 - frac))
    re_shift = None
    if "re_logits" in taps:                                  # keep the rescoring branch selective too
        r = taps["re_logits"].mean(-2).flatten()
        re_shift = logit_thr - float(torch.quantile(r, 1 - frac * 0.6))
    return shift, re_shift


def case_e2e(builtin, tag, frames=8, hw=(96, 128)):
    """(iii)+(vi) whole path on tiny frames through the reference's unmodified inference/batch_inference/
    _remove_short_track/batch_postprocess (backbone = oracle R-50 restatement, unpinned)."""
    cfg = mini_cfg(builtin)
    sd = synth_state_dict(cfg, seed=7)
    clip = make_clip(frames, hw[0], hw[1], clip_id=1)
    images = [torch.as_tensor(f.astype("float32").transpose(2, 0, 1)) for f in clip]
    shift, re_shift = calibrate(cfg, sd, images)
    key = "detection_transformer.ctrl_point_class.0.bias"
    bias = float(sd[key][0]) + shift
    cls_bias = {key: bias}
    re_bias = 0.0
    if re_shift is not None:
        re_bias = float(sd["roi_heads.rescoring_head.bias"][0]) + re_shift
        cls_bias["roi_heads.rescoring_head.bias"] = re_bias
    sd = synth_state_dict(cfg, seed=7, cls_bias=cls_bias)
    model = build_ref_gomatching(cfg, sd)
    inputs = [{"image": im, "height": hw[0], "width": hw[1], "video_id": 0} for im in images]
    tc = new_time_cost()
    with torch.no_grad():
        insts, id_count = model.batch_inference(copy.deepcopy(inputs), 0, 0, [], tc)
        ids_before = [x.track_ids.clone() for x in insts]
        pre = [{k: (v.tensor.clone() if hasattr(v, "tensor") else v.clone()) for k, v in x._fields.items()}
               for x in insts]
        insts = model._remove_short_track(insts)
        res = model.batch_postprocess(insts, [hw] * len(insts))
        mine, mc = O.run_clip(sd, cfg, [im.clone() for im in images])
    out = {"cls_bias": np.asarray([bias, re_bias], dtype=np.float32), "hw": np.asarray(hw),
           "num_frames": np.asarray([frames]), "id_count": np.asarray([int(id_count)])}
    worst = 0.0
    same_ids = int(mc) == int(id_count)
    for t in range(frames):
        out["pre_ids_%d" % t] = _np(ids_before[t])
        out["pre_scores_%d" % t] = _np(pre[t]["scores"])
        out["pre_boxes_%d" % t] = _np(pre[t]["pred_boxes"])
        r = res[t]["instances"]
        m = mine[t]["instances"]
        for k in ("track_ids", "scores", "recs", "bd", "ctrl_points"):
            out["%s_%d" % (k, t)] = _np(getattr(r, k))
        out["pred_boxes_%d" % t] = _np(r.pred_boxes.tensor)
        same_ids &= torch.equal(r.track_ids, m["track_ids"]) and torch.equal(r.recs, m["recs"])
        if len(r):
            worst = max(worst, _maxdiff(r.bd, m["bd"]), _maxdiff(r.ctrl_points, m["ctrl_points"]),
                        _maxdiff(r.scores, m["scores"]))
    print("e2e/%s dets/frame %s kept %s id_count %d | oracle ids/recs identical: %s, max|d| pts/scores = %.3e" % (
        tag, [len(x) for x in ids_before], [len(r["instances"]) for r in res], int(id_count), same_ids, worst))
    np.savez_compressed(os.path.join(GOLD, "e2e_%s.npz" % tag), **out)


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    case_msda()
    case_deepsolo("icdar15", "ic15")
    case_deepsolo("bovtext", "voc96", voc=96)
    case_deepsolo("icdar15", "padded", image_hw=(41, 70))
    case_matcher("icdar15", "lst")
    case_matcher("pp_dstext", "pp")
    case_tracker("icdar15", "lst")
    case_tracker("pp_dstext", "pp")
    case_e2e("icdar15", "lst")
    case_e2e("pp_dstext", "pp")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "padded":           # regenerate only the padded-batch DeepSolo fixture
        os.makedirs(GOLD, exist_ok=True)
        torch.manual_seed(0)
        torch.set_num_threads(8)
        case_deepsolo("icdar15", "padded", image_hw=(41, 70))
    else:
        main()
