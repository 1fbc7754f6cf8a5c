"""GPU parity of the assembled path (backbone, DeepSolo, matcher heads, tracker, whole clip) against the
reference-generated fixtures and the CPU oracle."""
import numpy as np
import pytest
import torch

from helpers import mini_cfg, golden, e2e_state_dict, t

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(params=["f16x3", "bf16x6", "fp32"])
def gemm_mode(request):
    """All contraction back-ends -- two-plane fp16 split (default), three-plane bf16 split, exact-fp32 MFMA -- held to the
    same tolerances."""
    from gomatching_amd import ops
    old = ops.GEMM_MODE
    ops.GEMM_MODE = request.param
    yield request.param
    ops.GEMM_MODE = old


def _close(a, b, atol, msg=""):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (msg, a.shape, b.shape)
    err = (a - b).abs()
    assert bool((err <= atol).all()), "%s max|d|=%.3e (tol %.1e)" % (msg, float(err.max()) if err.numel() else 0, atol)


def _time_cost():
    return {k: 0.0 for k in ("pre_process", "backbone", "detector", "rescore", "tracker", "short_match",
                             "long_match", "post_process", "total_time")}


def test_backbone_vs_oracle(gemm_mode):
    from gomatching_amd.weights import synth_state_dict
    from gomatching_amd.modeling import ResNet50
    from gomatching_amd import ops
    from oracle import gom_oracle as O
    cfg = mini_cfg()
    sd = synth_state_dict(cfg, seed=7)
    g = torch.Generator().manual_seed(2)
    img = torch.rand(2, 3, 96, 128, generator=g) * 255
    mean = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(cfg.MODEL.PIXEL_STD).view(1, 3, 1, 1)
    with torch.no_grad():
        ref = O.resnet50((img - mean) / std, sd)
    net = ResNet50(sd, DEV)
    out = net.forward(ops.preprocess(img.to(DEV), cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD))
    for k in ("res3", "res4", "res5"):
        scale = float(ref[k].abs().max())
        _close(out[k].permute(0, 3, 1, 2), ref[k], 2e-5 * max(scale, 1.0), k)


def test_msda_window_policy_follows_the_measured_offsets():
    """ops.MSDA_WINDOW_POLICY: every encoder layer decides on its first eager call, from the counted share of octet groups whose
    samples leave the 5-pixel windows, whether its level-0 queries run on the LDS-window kernel.  With the synthetic weights' offsets
    nothing falls back and the windows stay; with the sampling offsets scaled x 8 (a stand-in for a trained checkpoint's larger
    offsets) the layers switch to the gather kernel.  The two kernels are bit-identical, so the outputs do not depend on the
    choice: checked against a run with the policy off."""
    from gomatching_amd import ops
    from gomatching_amd.weights import synth_state_dict
    from gomatching_amd.modeling import DeepSolo
    cfg = mini_cfg("icdar15")
    gen = torch.Generator().manual_seed(3)
    # maps large enough for a window to be a PART of them (on the mini fixtures' 8 x 12 maps every window is the whole map)
    feats = [(torch.randn((1, h, w, c), generator=gen) * 0.5).to(DEV) for (h, w), c in zip(((64, 96), (32, 48), (16, 24)), (512, 1024, 2048))]
    old = ops.MSDA_WINDOW_POLICY
    try:
        with ops.gemm_mode("f16x3"):
            for scale, want in ((1.0, True), (8.0, False)):
                sd = dict(synth_state_dict(cfg, seed=7))
                for k in list(sd):
                    if ".encoder.layers." in k and "sampling_offsets" in k:
                        sd[k] = sd[k] * scale
                ops.MSDA_WINDOW_POLICY = True
                net = DeepSolo(cfg, sd, DEV)
                out = net.forward(feats)
                assert sorted(net.msda_window_fallback) == list(range(net.n_enc))
                assert all(L["msda_window"] is want for L in net.enc), (scale, net.msda_window_fallback)
                if want:
                    assert max(net.msda_window_fallback.values()) < 0.01   # (a handful of octet groups on random features)
                else:
                    assert min(net.msda_window_fallback.values()) > ops.MSDA_WINDOW_MAX_FALLBACK
                ops.MSDA_WINDOW_POLICY = False
                net2 = DeepSolo(cfg, sd, DEV)
                out2 = net2.forward(feats)
                assert net2.msda_window_fallback == {}
                for k in ("pred_logits", "pred_ctrl_points", "query_features"):
                    assert torch.equal(out[k], out2[k]), (scale, k)
    finally:
        ops.MSDA_WINDOW_POLICY = old


@pytest.mark.parametrize("builtin,tag,voc", [("icdar15", "ic15", None), ("bovtext", "voc96", 96)])
def test_deepsolo_mini_golden(builtin, tag, voc, gemm_mode):
    """DeepSolo-without-backbone against the reference's own outputs (mini geometry, B=2)."""
    from gomatching_amd.weights import synth_state_dict
    from gomatching_amd.modeling import DeepSolo
    g = golden("deepsolo_%s.npz" % tag)
    cfg = mini_cfg(builtin, voc=voc)
    sd = synth_state_dict(cfg, seed=7)
    net = DeepSolo(cfg, sd, DEV)
    feats = [t(g["feat%d" % i]).permute(0, 2, 3, 1).contiguous().to(DEV) for i in range(3)]
    taps = {}
    out = net.forward(feats, taps=taps)
    B, nq, P = 2, cfg.MODEL.TRANSFORMER.NUM_QUERIES, 25
    _close(taps["memory"].view(B, -1, 256), g["tap_memory"], 1e-4, "memory")
    assert torch.equal(taps["topk"].cpu().long(), t(g["tap_topk"])), "top-k proposals differ"
    _close(taps["init_ref"].view(B, nq, P, 2), g["tap_init_ref"], 1e-5, "init_ref")
    for k, shape in (("pred_logits", (B, nq, P, 1)), ("pred_text_logits", (B, nq, P, -1)),
                     ("pred_ctrl_points", (B, nq, P, 2)), ("pred_bd_points", (B, nq, P, 4)),
                     ("query_features", (B, nq, P, 256))):
        _close(out[k].view(*shape), g[k], 2e-4, k)


def test_deepsolo_padded_batch_golden(gemm_mode):
    """Padded batch (41x70 images inside 64x96): the mask-aware kernels against the reference's own outputs."""
    from gomatching_amd.weights import synth_state_dict
    from gomatching_amd.modeling import DeepSolo
    g = golden("deepsolo_padded.npz")
    cfg = mini_cfg("icdar15")
    sd = synth_state_dict(cfg, seed=7)
    net = DeepSolo(cfg, sd, DEV)
    feats = [t(g["feat%d" % i]).permute(0, 2, 3, 1).contiguous().to(DEV) for i in range(3)]
    taps = {}
    out = net.forward(feats, taps=taps, image_hw=tuple(int(v) for v in g["image_hw"]))
    B, nq, P = 2, cfg.MODEL.TRANSFORMER.NUM_QUERIES, 25
    # padded tokens of the memory are garbage-in/garbage-out in both implementations only through masked paths;
    # compare the valid ones, then everything downstream
    shapes = [(8, 12), (4, 6), (2, 3), (1, 2)]
    vs = DeepSolo.valid_shapes(shapes, tuple(int(v) for v in g["image_hw"]))
    keep = torch.cat([((torch.arange(H)[:, None] < v[0]) & (torch.arange(W)[None, :] < v[1])).flatten()
                      for (H, W), v in zip(shapes, vs)])
    mem = taps["memory"].view(B, -1, 256).cpu()
    _close(mem[:, keep], t(g["tap_memory"])[:, keep], 1e-4, "memory (valid tokens)")
    assert torch.equal(taps["topk"].cpu().long(), t(g["tap_topk"])), "top-k proposals differ"
    _close(taps["init_ref"].view(B, nq, P, 2), g["tap_init_ref"], 1e-5, "init_ref")
    for k, shape in (("pred_logits", (B, nq, P, 1)), ("pred_text_logits", (B, nq, P, -1)),
                     ("pred_ctrl_points", (B, nq, P, 2)), ("pred_bd_points", (B, nq, P, 4)),
                     ("query_features", (B, nq, P, 256))):
        _close(out[k].view(*shape), g[k], 2e-4, k)


@pytest.mark.parametrize("builtin,tag", [("icdar15", "lst"), ("pp_dstext", "pp")])
def test_matcher_heads_golden(builtin, tag):
    from gomatching_amd.weights import synth_state_dict
    from gomatching_amd.modeling import build_roi_heads
    g = golden("matcher_%s.npz" % tag)
    cfg = mini_cfg(builtin)
    sd = synth_state_dict(cfg, seed=7)
    rh = build_roi_heads(cfg, sd, DEV)
    for n in (1, 7, 20):
        x = t(g["fc_in_%d" % n]).reshape(n, -1).to(DEV)
        rows = torch.arange(n, dtype=torch.int32, device=DEV)
        _close(rh.asso_head(x, rows), g["fc_out_%d" % n], 1e-4, "fchead")
    for ci in range(5):
        n_t = [int(v) for v in g["asso%d_nt" % ci]]
        k, short = int(g["asso%d_k" % ci][0]), bool(g["asso%d_k" % ci][1])
        if n_t[k] == 0:
            continue
        reid = t(g["asso%d_reid" % ci]).to(DEV)
        logits = rh._forward_transformer(reid, n_t, k, short_term=short)
        _close(logits, g["asso%d_logits" % ci], 2e-3, "asso logits")
        _close(rh._activate_asso(logits, n_t), g["asso%d_out" % ci], 1e-4, "asso act")


@pytest.fixture(params=["tracker_rt", "native", "python"])
def matcher_runtime(request):
    """tracker_rt: the whole per-frame recurrence in native code (default).  Otherwise the Python loop of `track_frames`
    with the per-match device chain issued kernel by kernel by the native runtime (one FFI call per match), or composed kernel
    by kernel in Python."""
    from gomatching_amd import ops
    old = ops.NATIVE_MATCHER, ops.NATIVE_TRACKER
    ops.NATIVE_TRACKER = request.param == "tracker_rt"
    ops.NATIVE_MATCHER = request.param != "python"
    yield request.param
    ops.NATIVE_MATCHER, ops.NATIVE_TRACKER = old


@pytest.mark.parametrize("builtin,tag", [("icdar15", "lst"), ("pp_dstext", "pp")])
def test_tracker_trace_golden(builtin, tag, matcher_runtime):
    """The reference's 16-frame id trace (empty frame, births, drop-outs, long-term re-association)."""
    from gomatching_amd.weights import synth_state_dict
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.structures import Instances, Boxes
    g = golden("tracker_%s.npz" % tag)
    cfg = mini_cfg(builtin, device=DEV)
    sd = synth_state_dict(cfg, seed=7)
    model = GoMatching(cfg, sd, device=DEV)
    size = tuple(int(v) for v in g["image_size"])
    frames = int(g["num_frames"][0])
    dets = []
    for f in range(frames):
        inst = Instances(size)
        inst.reid_features = t(g["reid_%d" % f]).to(DEV)
        inst.pred_boxes = Boxes(t(g["boxes_%d" % f]).to(DEV))
        dets.append(inst)
    it = iter(dets)
    model.detect_launch = lambda batched_inputs, time_cost: list(batched_inputs)      # detection stubbed out:
    model.detect_finish = lambda h, time_cost: [next(it) for _ in h]                   # the reference's fixtures
    insts, id_count = model.batch_inference([{} for _ in range(frames)], 0, 0, [], _time_cost())
    assert int(id_count) == int(g["id_count"][0])
    for f in range(frames):
        assert insts[f].track_ids.cpu().tolist() == g["ids_%d" % f].tolist(), f
    kept = model._remove_short_track(insts)
    for f in range(frames):
        assert kept[f].track_ids.cpu().tolist() == g["kept_ids_%d" % f].tolist(), f


def _synthetic_trace(frames, dim, seed):
    """Moving objects with births, drop-outs and re-appearances: per frame (reid [n, dim], boxes [n, 4])."""
    g = np.random.default_rng(seed)
    objs = [{"f": g.standard_normal(dim).astype(np.float32), "xy": g.uniform(10, 90, 2), "v": g.uniform(-0.4, 0.4, 2),
             "on": True} for _ in range(5)]
    out = []
    for t in range(frames):
        if g.random() < 0.06 and len(objs) < 9:
            objs.append({"f": g.standard_normal(dim).astype(np.float32), "xy": g.uniform(10, 90, 2),
                         "v": g.uniform(-0.4, 0.4, 2), "on": True})
        feats, boxes = [], []
        for o in objs:
            o["xy"] = np.clip(o["xy"] + o["v"], 2, 100)
            if g.random() < 0.08:
                o["on"] = not o["on"]                                # drop-out / re-appearance
            if o["on"] and not (t % 37 == 20):                      # and a few entirely empty frames
                feats.append(o["f"] + 0.15 * g.standard_normal(dim).astype(np.float32))
                boxes.append([o["xy"][0], o["xy"][1], o["xy"][0] + 14, o["xy"][1] + 8])
        out.append((np.asarray(feats, np.float32).reshape(-1, dim), np.asarray(boxes, np.float32).reshape(-1, 4)))
    return out


@pytest.mark.parametrize("builtin", ["icdar15", "pp_dstext"])
def test_tracker_across_100_frame_batches_vs_oracle(builtin, matcher_runtime):
    """eval.py feeds a video in 100-frame batches and carries (instances, id_count) (eval.py:326-344,
    gom_lstmatcher.py:366-403 with start_frame_id = batch_id * 100): ids of a 106-frame trace processed as 100 + 6
    frames equal the oracle's, before and after short-track removal -- the window, the embedding pool and the id
    counter all survive the batch boundary."""
    from oracle import gom_oracle as O
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.structures import Instances, Boxes
    from gomatching_amd.weights import synth_state_dict
    cfg = mini_cfg(builtin, device=DEV)
    sd = synth_state_dict(cfg, seed=7)
    model = GoMatching(cfg, sd, device=DEV)
    size = (96, 128)
    trace = _synthetic_trace(106, model.roi_heads.feature_dim, seed=5)
    ocfg = mini_cfg(builtin)
    o_insts = [O.Inst(size, reid_features=torch.from_numpy(f).clone(), pred_boxes=torch.from_numpy(b).clone())
               for f, b in trace]
    with torch.no_grad():
        o_res, o_count = O.track_clip(sd, ocfg, o_insts[:100], batch_id=0)
        o_res, o_count = O.track_clip(sd, ocfg, o_insts[100:], batch_id=1, id_count=o_count, instances=o_res)
    dets = []
    for f, b in trace:
        inst = Instances(size)
        inst.reid_features = torch.from_numpy(f).to(DEV)
        inst.pred_boxes = Boxes(torch.from_numpy(b).to(DEV))
        dets.append(inst)
    it = iter(dets)
    model.detect_launch = lambda batched_inputs, time_cost: list(batched_inputs)
    model.detect_finish = lambda h, time_cost: [next(it) for _ in h]
    insts, id_count = model.batch_inference([{} for _ in range(100)], 0, 0, [], _time_cost())
    insts, id_count = model.batch_inference([{} for _ in range(6)], 1, id_count, insts, _time_cost())
    assert len(insts) == 106 and int(id_count) == int(o_count)
    for f in range(106):
        assert insts[f].track_ids.cpu().tolist() == o_res[f]["track_ids"].tolist(), f
    kept, o_kept = model._remove_short_track(insts), O.remove_short_track(ocfg, o_res)
    for f in range(106):
        assert kept[f].track_ids.cpu().tolist() == o_kept[f]["track_ids"].tolist(), f
    assert max(max(x.track_ids.cpu().tolist(), default=0) for x in kept) > 5       # births happened


@pytest.mark.parametrize("builtin", ["icdar15", "pp_dstext"])
def test_tracker_mixed_resolution_vs_oracle(builtin, matcher_runtime):
    """A clip whose frames change size (BASELINE config #5): a short-term match divides each frame's boxes by its OWN image size
    (lstmatcher.py:478-494), a long-term match divides every frame of its window by the size of the window's FIRST frame
    (gom_lstmatcher.py:471) -- ids equal the oracle's on a 40-frame trace that switches between three sizes, for all runtimes."""
    from oracle import gom_oracle as O
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.structures import Instances, Boxes
    from gomatching_amd.weights import synth_state_dict
    cfg = mini_cfg(builtin, device=DEV)
    sd = synth_state_dict(cfg, seed=7)
    model = GoMatching(cfg, sd, device=DEV)
    # objects that move slowly and carry NO appearance cue (fresh random embeddings every frame): the association falls to the
    # IoU term, i.e. to how the boxes are normalised
    g = np.random.default_rng(11)
    objs = [{"xy": g.uniform(10, 90, 2), "v": g.uniform(-0.4, 0.4, 2)} for _ in range(6)]
    trace = []
    for _ in range(40):
        feats, boxes = [], []
        for o in objs:
            o["xy"] = np.clip(o["xy"] + o["v"], 2, 100)
            if g.random() < 0.9:
                feats.append(g.standard_normal(model.roi_heads.feature_dim).astype(np.float32))
                boxes.append([o["xy"][0], o["xy"][1] * 0.7, o["xy"][0] + 14, o["xy"][1] * 0.7 + 8])
        trace.append((np.asarray(feats, np.float32).reshape(-1, model.roi_heads.feature_dim), np.asarray(boxes, np.float32).reshape(-1, 4)))
    sizes = [((96, 128), (128, 96), (192, 256))[(f // 4) % 3] for f in range(40)]
    scaled = [(f_, b * np.array([sz[1] / 128.0, sz[0] / 96.0, sz[1] / 128.0, sz[0] / 96.0], np.float32))
              for (f_, b), sz in zip(trace, sizes)]
    ocfg = mini_cfg(builtin)
    o_insts = [O.Inst(sz, reid_features=torch.from_numpy(f_).clone(), pred_boxes=torch.from_numpy(b).clone())
               for (f_, b), sz in zip(scaled, sizes)]
    with torch.no_grad():
        o_res, o_count = O.track_clip(sd, ocfg, o_insts)
    dets = []
    for (f_, b), sz in zip(scaled, sizes):
        inst = Instances(sz)
        inst.reid_features = torch.from_numpy(f_).to(DEV)
        inst.pred_boxes = Boxes(torch.from_numpy(b).to(DEV))
        dets.append(inst)
    it = iter(dets)
    model.detect_launch = lambda batched_inputs, time_cost: list(batched_inputs)
    model.detect_finish = lambda h, time_cost: [next(it) for _ in h]
    insts, id_count = model.batch_inference([{} for _ in range(40)], 0, 0, [], _time_cost())
    assert int(id_count) == int(o_count)
    for f in range(40):
        assert insts[f].track_ids.cpu().tolist() == o_res[f]["track_ids"].tolist(), f
    # the sizes matter: normalising everything by one size gives other ids on this trace (the test would not notice otherwise)
    same = [O.Inst(sizes[0], reid_features=torch.from_numpy(f_).clone(), pred_boxes=torch.from_numpy(b).clone()) for f_, b in scaled]
    with torch.no_grad():
        s_res, _ = O.track_clip(sd, ocfg, same)
    assert any(a["track_ids"].tolist() != b["track_ids"].tolist() for a, b in zip(s_res, o_res))


@pytest.mark.parametrize("builtin", ["icdar15", "pp_dstext"])
@pytest.mark.parametrize("n_t,k", [([7, 0, 12, 5], 3), ([60, 70, 90], 2), ([1, 1], 1), ([3, 140], 1)])
def test_native_match_runtime_equals_python_composition(builtin, n_t, k):
    """gom_match_scores_f32 (matcher_rt.cpp) returns the same bits as the per-kernel Python composition, for the
    long- and the short-term matcher, rows below and above the 64-row kernel switch, an empty frame in the window."""
    from gomatching_amd import ops
    from gomatching_amd.modeling.roi_heads import build_roi_heads
    from gomatching_amd.weights import synth_state_dict
    cfg = mini_cfg(builtin, device=DEV)
    heads = build_roi_heads(cfg, synth_state_dict(cfg, seed=11), torch.device(DEV))
    g = torch.Generator().manual_seed(sum(n_t) + k)
    N, T, n_k = sum(n_t), len(n_t), n_t[k]
    pool = torch.randn(N + 9, heads.feature_dim, generator=g).to(DEV)
    rows = torch.randperm(N + 9, generator=g)[:N].to(torch.int32).to(DEV)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(n_t)]), dtype=torch.int32, device=DEV)
    Np = N - n_k
    lo = sum(n_t[:k])
    M = max(1, Np // 2)
    col_of = np.arange(Np) % M
    last = np.array([np.nonzero(col_of == m)[0].max() for m in range(M)])
    nonk = np.concatenate([np.arange(0, lo), np.arange(lo + n_k, N)])
    meta = torch.tensor(np.concatenate([nonk, col_of, last, np.arange(lo, lo + n_k)]), dtype=torch.int32, device=DEV)
    xy = torch.rand(N, 2, generator=g) * 80
    boxes = torch.cat([xy, xy + 10 + torch.rand(N, 2, generator=g) * 30], 1).to(DEV)
    decay = (0.9 ** torch.arange(Np).float()).to(DEV)
    out = {}
    for short_term in (False, True):
        for mode in ("native", "python"):
            ops.NATIVE_MATCHER = mode != "python"
            try:
                out[mode] = heads.match_scores(pool, rows, offs, meta, boxes, None if short_term else decay, n_t, k,
                                               short_term, (96, 128), M, True, 0.0 if short_term else 50.0)
            finally:
                ops.NATIVE_MATCHER = True
        assert out["native"].shape == (n_k, M) and torch.isfinite(out["native"]).all()
        assert torch.equal(out["native"], out["python"])


@pytest.mark.parametrize("builtin", ["icdar15", "pp_dstext"])
def test_batched_short_term_equals_per_pair(builtin):
    """All frame pairs in one ragged launch per op (segmented attention + fused logits/softmax/IoU kernel) give the same
    scores as the per-pair kernels: ragged sizes, a single-detection frame, > 64 detections in a frame."""
    from gomatching_amd import ops
    from gomatching_amd.modeling.roi_heads import build_roi_heads
    from gomatching_amd.weights import synth_state_dict
    cfg = mini_cfg(builtin, device=DEV)
    heads = build_roi_heads(cfg, synth_state_dict(cfg, seed=11), torch.device(DEV))
    g = torch.Generator().manual_seed(3)
    sizes = [(5, 9), (9, 1), (1, 70), (70, 33), (33, 33)]
    pairs, off = [], 0
    for n_prev, n_cur in sizes:
        pairs.append((off, n_prev, n_cur))
        off += n_prev + n_cur
    src = torch.randn(off, heads.feature_dim, generator=g).to(DEV)
    xy = torch.rand(off, 2, generator=g) * 80
    boxes = torch.cat([xy, xy + 10 + torch.rand(off, 2, generator=g) * 30], 1).to(DEV)
    res = {}
    for batched in (True, False):
        ops.BATCHED_SHORT_TERM = batched
        try:
            res[batched] = [x.clone() for x in heads.short_term_scores(src, pairs, boxes, (96, 128))]
        finally:
            ops.BATCHED_SHORT_TERM = True
    for (_, n_prev, n_cur), a, b in zip(pairs, res[True], res[False]):
        assert a.shape == b.shape == (n_cur, n_prev)
        assert torch.isfinite(a).all() and float(a.min()) >= 0.0 and float(a.max()) <= 1.0
        _close(a, b.cpu(), 1e-6, "batched vs per-pair short-term scores")


@pytest.mark.parametrize("builtin,tag", [("icdar15", "lst"), ("pp_dstext", "pp")])
@pytest.mark.parametrize("step", [8, 3])
def test_end_to_end_clip_golden(builtin, tag, step, gemm_mode):
    """Whole path on the 8-frame mini clip against the reference's outputs: identical ids and characters,
    points within 1e-3 px."""
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.synth import make_clip
    g = golden("e2e_%s.npz" % tag)
    cfg = mini_cfg(builtin, device=DEV)
    sd = e2e_state_dict(cfg, g)
    hw = tuple(int(v) for v in g["hw"])
    frames = int(g["num_frames"][0])
    clip = make_clip(frames, hw[0], hw[1], clip_id=1)
    inputs = [{"image": torch.as_tensor(f.astype("float32").transpose(2, 0, 1)), "height": hw[0], "width": hw[1]}
              for f in clip]
    model = GoMatching(cfg, sd, device=DEV, frames_per_step=step)
    insts, id_count = model.batch_inference(inputs, 0, 0, [], _time_cost())
    for f in range(frames):
        assert insts[f].track_ids.cpu().tolist() == g["pre_ids_%d" % f].tolist(), ("pre ids", f)
        _close(insts[f].scores, g["pre_scores_%d" % f], 1e-4, "pre scores")
        _close(insts[f].pred_boxes.tensor, g["pre_boxes_%d" % f], 1e-3, "pre boxes")
    assert int(id_count) == int(g["id_count"][0])
    insts = model._remove_short_track(insts)
    res = model.batch_postprocess(insts, [hw] * len(insts))
    for f in range(frames):
        r = res[f]["instances"]
        assert r.track_ids.cpu().tolist() == g["track_ids_%d" % f].tolist()
        assert r.recs.cpu().tolist() == g["recs_%d" % f].tolist()
        _close(r.scores, g["scores_%d" % f], 1e-4, "scores")
        _close(r.bd, g["bd_%d" % f], 1e-3, "bd")
        _close(r.ctrl_points, g["ctrl_points_%d" % f], 1e-3, "ctrl")
        _close(r.pred_boxes.tensor, g["pred_boxes_%d" % f], 1e-3, "boxes")


def test_batch_invariance_and_determinism():
    """Size-independent properties at a larger geometry: a frame's detections do not depend on what shares its
    step, and the path is run-to-run deterministic."""
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.synth import make_clip
    from gomatching_amd.weights import synth_state_dict
    cfg = mini_cfg("icdar15", nq=100, device=DEV)
    sd = synth_state_dict(cfg, seed=3, cls_bias={"detection_transformer.ctrl_point_class.0.bias": 0.0})
    clip = make_clip(3, 320, 480, clip_id=5)
    inputs = [{"image": torch.as_tensor(f.astype("float32").transpose(2, 0, 1))} for f in clip]
    model = GoMatching(cfg, sd, device=DEV)
    a = model.inference(inputs, _time_cost())
    b = model.inference(inputs[1:2], _time_cost())
    c = model.inference(inputs, _time_cost())
    assert len(a[1]) == len(b[0]) and len(a[1]) > 0
    assert torch.equal(a[1].scores, b[0].scores) and torch.equal(a[1].bd, b[0].bd)
    assert torch.equal(a[1].reid_features, b[0].reid_features)
    for x, y in zip(a, c):
        assert torch.equal(x.scores, y.scores) and torch.equal(x.recs, y.recs) and torch.equal(x.bd, y.bd)


@pytest.mark.parametrize("builtin,hw,nframes", [("pp_dstext", (1280, 2276), 1), ("bovtext", (1000, 1778), 2)])
def test_full_size_configs_run(builtin, hw, nframes):
    """BASELINE configs #4 (300 queries, 1280x2276, GoMatching++) and #5 (voc 5462) at their full sizes: the path
    runs, outputs are finite and in range, and the result is run-to-run identical (size-independent properties)."""
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.config import setup_cfg
    from gomatching_amd.weights import synth_state_dict
    cfg = setup_cfg(builtin=builtin)
    cfg.MODEL.DEVICE = DEV
    sd = synth_state_dict(cfg, seed=1, cls_bias={"detection_transformer.ctrl_point_class.0.bias": 0.3})
    model = GoMatching(cfg, sd, device=DEV, frames_per_step=2)
    g = torch.Generator().manual_seed(3)
    inputs = [{"image": torch.rand(3, hw[0], hw[1], generator=g) * 255} for _ in range(nframes)]
    a = model.inference(inputs, _time_cost())
    b = model.inference(inputs, _time_cost())
    nq, voc = cfg.MODEL.TRANSFORMER.NUM_QUERIES, cfg.MODEL.TRANSFORMER.VOC_SIZE
    for x, y in zip(a, b):
        n = len(x)
        assert 0 < n <= nq
        assert torch.isfinite(x.bd).all() and torch.isfinite(x.reid_features).all()
        assert float(x.scores.min()) > cfg.MODEL.TRANSFORMER.INFERENCE_TH_TEST and float(x.scores.max()) <= 1.0
        assert int(x.recs.min()) >= 0 and int(x.recs.max()) <= voc
        assert float(x.bd[..., 0::2].max()) <= hw[1] + 1e-3 and float(x.bd[..., 1::2].max()) <= hw[0] + 1e-3
        assert torch.equal(x.scores, y.scores) and torch.equal(x.recs, y.recs) and torch.equal(x.bd, y.bd)
        s = x.scores.cpu()
        assert bool((s[:-1] >= s[1:]).all())                    # NMS order = descending score


def test_detector_graph_replay_equals_eager_and_is_bounded():
    """hipGraph replay of the detector returns the bits of the eager launches, survives interleaved step shapes, and
    keeps at most `max_graphs` captured shapes alive."""
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.weights import synth_state_dict
    cfg = mini_cfg("icdar15", device=DEV)
    sd = synth_state_dict(cfg, seed=7, cls_bias={"detection_transformer.ctrl_point_class.0.bias": 0.5})
    g = torch.Generator().manual_seed(4)
    shapes = [(96, 128), (128, 96), (64, 160)]
    clips = {hw: [{"image": (torch.rand(3, hw[0], hw[1], generator=g) * 255).to(DEV)} for _ in range(2)] for hw in shapes}
    eager = GoMatching(cfg, sd, device=DEV, use_graphs=False)
    ref = {hw: eager.inference(clips[hw], _time_cost()) for hw in shapes}
    model = GoMatching(cfg, sd, device=DEV, use_graphs=True)
    for rnd in range(4):                                             # eager, capture, replay, replay -- interleaved
        for hw in shapes:
            got = model.inference(clips[hw], _time_cost())
            for a, b in zip(got, ref[hw]):
                assert torch.equal(a.scores, b.scores) and torch.equal(a.bd, b.bd) and torch.equal(a.recs, b.recs)
                assert a.has("reid_features") == b.has("reid_features")
            if a.has("reid_features"):
                assert torch.equal(a.reid_features, b.reid_features)
    assert model.use_graphs                                          # capture did not fall back
    assert sum(isinstance(v, dict) for v in model._graphs.values()) <= model.max_graphs


def test_mixed_resolution_clip():
    """Frames of different sizes in one batch_inference call are split into per-size steps (config #5)."""
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.weights import synth_state_dict
    cfg = mini_cfg("icdar15", device=DEV)
    sd = synth_state_dict(cfg, seed=7, cls_bias={"detection_transformer.ctrl_point_class.0.bias": 0.5})
    model = GoMatching(cfg, sd, device=DEV, frames_per_step=4)
    g = torch.Generator().manual_seed(9)
    sizes = [(96, 128), (96, 128), (128, 96), (96, 128), (96, 128)]
    inputs = [{"image": torch.rand(3, h, w, generator=g) * 255} for h, w in sizes]
    insts, id_count = model.batch_inference(inputs, 0, 0, [], _time_cost())
    assert [x.image_size for x in insts] == sizes
    single = [model.inference([x], _time_cost())[0] for x in inputs]
    for a, b in zip(insts, single):
        assert torch.equal(a.scores, b.scores) and torch.equal(a.bd, b.bd)
    for x in insts:
        ids = x.track_ids.cpu().tolist()
        assert len(ids) == len(set(ids))


def test_results_survive_next_video_and_tracker_handle_follows_thresholds():
    """(a) What `batch_postprocess` hands back must not alias the embedding pool: the next video's `begin_batch` re-uses
    the pool from row 0 (in the reference every frame owns its tensor).  (b) The native tracker handle caches thresholds
    and weight pointers at creation: changing a threshold makes a new handle (the old one is destroyed), the ids then
    follow the NEW threshold exactly as the Python loop (which reads the attributes every call) does."""
    from gomatching_amd import ops
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.structures import Instances, Boxes
    from gomatching_amd.weights import synth_state_dict
    cfg = mini_cfg("icdar15", device=DEV)
    model = GoMatching(cfg, synth_state_dict(cfg, seed=7), device=DEV)
    size = (96, 128)

    def run(trace, native=True):
        old, ops.NATIVE_TRACKER = ops.NATIVE_TRACKER, native
        try:
            dets = []
            for f, b in trace:
                inst = Instances(size)
                inst.reid_features = torch.from_numpy(f).to(DEV)
                inst.pred_boxes = Boxes(torch.from_numpy(b).to(DEV))
                inst.bd = torch.zeros((len(b), 4), device=DEV)            # batch_postprocess scales these two fields
                inst.ctrl_points = torch.zeros((len(b), 2), device=DEV)
                dets.append(inst)
            it = iter(dets)
            model.detect_launch = lambda batched_inputs, time_cost: list(batched_inputs)
            model.detect_finish = lambda h, time_cost: [next(it) for _ in h]
            insts, _ = model.batch_inference([{} for _ in trace], 0, 0, [], _time_cost())
            return model.batch_postprocess(insts, [size] * len(insts))
        finally:
            ops.NATIVE_TRACKER = old

    tr_a = _synthetic_trace(20, model.roi_heads.feature_dim, seed=5)
    tr_b = _synthetic_trace(20, model.roi_heads.feature_dim, seed=6)
    res_a = run(tr_a)
    kept = [(i, r["instances"].reid_features.clone()) for i, r in enumerate(res_a) if r["instances"].has("reid_features")]
    assert kept, "the last test_len frames carry their embeddings"
    h0 = model._ntrk
    run(tr_b)                                                          # second video: the pool is reused from row 0
    torch.cuda.synchronize()
    for i, f in kept:
        assert torch.equal(res_a[i]["instances"].reid_features, f), i
    assert model._ntrk == h0                                           # same settings: same handle
    ids_default = [r["instances"].track_ids.cpu().tolist() for r in run(tr_a)]
    model.overlap_thresh = 2.0                                         # no score reaches it: nothing associates any more
    ids_native = [r["instances"].track_ids.cpu().tolist() for r in run(tr_a)]
    assert model._ntrk_key[1] == 2.0
    ids_python = [r["instances"].track_ids.cpu().tolist() for r in run(tr_a, native=False)]
    assert ids_native == ids_python
    assert ids_native != ids_default
    flat = [i for f in ids_native for i in f]
    assert len(set(flat)) == len(flat)                                 # every detection is its own track
    model.close()
    assert model._ntrk is None
    model.close()                                                      # idempotent


def test_cu_partitioned_step_gives_identical_results():
    """`reserve_tracker_cus` (csrc/stream.hip): the detector on its own CU-masked stream, the tracker's per-frame recurrence on
    the complementary one.  A scheduling change only: detections, embeddings and ids of the 8-frame clip are identical bits /
    identical ids with and without the reservation, and undoing it works."""
    from gomatching_amd import ops
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.synth import make_clip
    g = golden("e2e_lst.npz")
    cfg = mini_cfg("icdar15", device=DEV)
    sd = e2e_state_dict(cfg, g)
    hw = tuple(int(v) for v in g["hw"])
    frames = int(g["num_frames"][0])
    clip = make_clip(frames, hw[0], hw[1], clip_id=1)
    inputs = [{"image": torch.as_tensor(f.astype("float32").transpose(2, 0, 1)), "height": hw[0], "width": hw[1]}
              for f in clip]
    model = GoMatching(cfg, sd, device=DEV, frames_per_step=3)

    def run():
        insts, id_count = model.batch_inference(inputs, 0, 0, [], _time_cost())
        torch.cuda.synchronize()
        return [(x.track_ids.cpu().tolist(), x.scores.clone(), x.pred_boxes.tensor.clone()) for x in insts], int(id_count)

    base, n0 = run()
    run()                                                               # graphs captured
    base2, _ = run()
    model.reserve_tracker_cus(32)
    assert model._lane_stream is not None and model._det_stream is not None
    for _ in range(3):                                                  # eager, capture, replay on the masked stream
        got, n1 = run()
        assert n1 == n0
        for (i0, s0, b0), (i1, s1, b1) in zip(base2, got):
            assert i0 == i1 and torch.equal(s0, s1) and torch.equal(b0, b1)
    for f in range(frames):
        assert got[f][0] == g["pre_ids_%d" % f].tolist()
    model.reserve_tracker_cus(0)
    assert model._lane_stream is None and model._det_stream is None
    again, _ = run()
    assert [a[0] for a in again] == [b[0] for b in base]
    lane = ops.masked_stream([0xFFFF, 0, 0, 0, 0, 0, 0, 0], torch.device(DEV, torch.cuda.current_device()))
    with torch.cuda.stream(lane):
        y = ops.gemm(torch.ones((8, 64), device=DEV), torch.ones((16, 64), device=DEV))
    lane.synchronize()
    assert float(y.min()) == 64.0 and float(y.max()) == 64.0


@pytest.mark.parametrize("builtin", ["icdar15", "pp_dstext"])
def test_hoisted_match_projections_change_nothing(builtin):
    """gom_match_scores_proj_f32 / gom_tracker_set_projections: the encoder in-projection and the decoder query projection of
    the raw embeddings computed once per detection instead of inside every match.  Same kernel, same bits: the 106-frame trace
    processed as 100 + 6 frames (pool restart, carried window, births, re-appearances) gives identical ids with and without,
    and the chain entry point gives identical trajectory scores."""
    from gomatching_amd import ops
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.structures import Instances, Boxes
    from gomatching_amd.weights import synth_state_dict
    cfg = mini_cfg(builtin, device=DEV)
    sd = synth_state_dict(cfg, seed=7)
    size = (96, 128)

    def run(hoist):
        old, ops.HOIST_MATCH_PROJECTIONS = ops.HOIST_MATCH_PROJECTIONS, hoist
        try:
            model = GoMatching(cfg, sd, device=DEV)
            trace = _synthetic_trace(106, model.roi_heads.feature_dim, seed=5)
            dets = []
            for f, b in trace:
                inst = Instances(size)
                inst.reid_features = torch.from_numpy(f).to(DEV)
                inst.pred_boxes = Boxes(torch.from_numpy(b).to(DEV))
                dets.append(inst)
            it = iter(dets)
            model.detect_launch = lambda batched_inputs, time_cost: list(batched_inputs)
            model.detect_finish = lambda h, time_cost: [next(it) for _ in h]
            insts, id_count = model.batch_inference([{} for _ in range(100)], 0, 0, [], _time_cost())
            insts, id_count = model.batch_inference([{} for _ in range(6)], 1, id_count, insts, _time_cost())
            return [x.track_ids.cpu().tolist() for x in insts], int(id_count)
        finally:
            ops.HOIST_MATCH_PROJECTIONS = old

    a, b = run(True), run(False)
    assert a == b and max(max(f, default=0) for f in a[0]) > 5


def test_precision_fallback_equals_pure_bf16x6():
    """A checkpoint whose activations leave fp16's range (here: res2.0.conv1's FrozenBN scaled by 2^18 and conv2's weight by
    2^-18 -- the same function, but the 3x3 convolution's A operand is beyond 65504, which used to be silently ZEROED behind
    its ReLU) must not raise and must not differ from a pure-bf16x6 run: every step whose range flag is up is re-run on the
    bf16x6 twin of the detector (eager step, captured step and replayed step alike); without the fallback it raises."""
    import warnings
    from gomatching_amd import ops
    from gomatching_amd.lib import GomError
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.synth import make_clip
    g = golden("e2e_lst.npz")
    cfg = mini_cfg("icdar15", device=DEV)
    sd = dict(e2e_state_dict(cfg, g))
    c = float(2 ** 18)
    p = "backbone.0.backbone.res2.0."
    sd[p + "conv1.norm.weight"], sd[p + "conv1.norm.bias"] = sd[p + "conv1.norm.weight"] * c, sd[p + "conv1.norm.bias"] * c
    sd[p + "conv2.weight"] = sd[p + "conv2.weight"] / c
    hw = tuple(int(v) for v in g["hw"])
    clip = make_clip(8, hw[0], hw[1], clip_id=1)
    inputs = [{"image": torch.as_tensor(f.astype("float32").transpose(2, 0, 1))} for f in clip]
    assert ops.GEMM_MODE == "f16x3"
    with ops.gemm_mode("bf16x6"):
        pure = GoMatching(cfg, sd, device=DEV, frames_per_step=4)
        ref, ref_idc = pure.batch_inference(inputs, 0, 0, [], _time_cost())
        assert pure.fallback_steps == 0
    assert sum(len(x) for x in ref) > 0
    assert [x.track_ids.cpu().tolist() for x in ref] == [g["pre_ids_%d" % f].tolist() for f in range(8)]   # power-of-two scaling
    model = GoMatching(cfg, sd, device=DEV, frames_per_step=4)
    for rnd in range(3):                                          # eager, capture, replay
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            insts, idc = model.batch_inference(inputs, 0, 0, [], _time_cost())
        assert any("fp16's range" in str(x.message) for x in w)
        assert model.fallback_steps == 2 * (rnd + 1)
        assert int(idc) == int(ref_idc)
        for a, b in zip(insts, ref):
            assert torch.equal(a.track_ids, b.track_ids) and torch.equal(a.scores, b.scores)
            assert torch.equal(a.recs, b.recs) and torch.equal(a.bd, b.bd) and torch.equal(a.ctrl_points, b.ctrl_points)
            assert a.has("reid_features") == b.has("reid_features")
            if a.has("reid_features"):
                assert torch.equal(a.reid_features, b.reid_features)
    model.precision_fallback = False
    with pytest.raises(GomError, match="fp16's range"):
        model.batch_inference(inputs, 0, 0, [], _time_cost())
    # in-range weights never take the fallback
    good = GoMatching(cfg, e2e_state_dict(cfg, g), device=DEV, frames_per_step=4)
    good.batch_inference(inputs, 0, 0, [], _time_cost())
    assert good.fallback_steps == 0 and good._fallback_det is None
