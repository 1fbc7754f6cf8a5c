"""GPU, slow: TRACKED clips of BASELINE configs #3, #4 and #5 at full size against the CPU oracle.

  #4  GoMatching++ / DSText: 8 frames 1920x1080 -> 1280x2276, 300 queries, SHA_FFN_CRSATTN, no rescoring, NMS 0.3, thr 0.5.
  #5  BOVText: voc 5462, ONE batch_inference call over frames of 1280x720, 1920x1080 and 720x1280 sources
      (-> 1000x1778, 1000x1778, 1778x1000 network inputs; the matcher normalises every frame's boxes by ITS size).
  #3  64 frames of 1280x720 as eight sequential 8-frame shards on ONE GPU -> pack_records -> concatenation ->
      unpack_records -> track_frames: ids equal to the single-process run and to the oracle's tracker.

How the oracle is used (kept under ~5 min per test): the CPU detector (oracle.detect_frames, ~10-25 s per frame on 32 threads)
runs on a SUBSET of the frames and must agree with the HIP detector there (same detections and characters, scores 1e-4,
points 1e-3 px, embeddings 1e-4); the oracle's TRACKER (track_clip + remove_short_track + batch_postprocess) then runs over
the HIP path's detections of ALL frames and must give identical ids.  GOM_FULL_ORACLE_CLIP=1 runs `oracle.run_clip` over
every frame instead (the log of such a run is committed under profiles/).  Tolerances: north_star (ids / characters identical,
points 1e-3 px).  Two facts bound what an fp32-vs-fp32 comparison can hold at these sizes, both measured:
  * the fp32 oracle itself sits 0.95e-3 px (4.2e-7 normalised) from its own float64 evaluation on the 1280x2276 frames
    (tools/diag/oracle_f64.py), and a coordinate near 2276 has an fp32 spacing of 2.4e-4 px: the point tolerance there is 8 ulp;
  * the proposal stage picks the top-300 of 60 640 class logits; ~25 of the 299 gaps between consecutive winners are below 1e-4
    and some below 1e-5 (tools/diag/topk_ties.py), i.e. inside the rounding noise of ANY fp32 evaluation (the reference's own
    included).  Where two near-tied winners swap RANK (a query slot = learned embedding + the token of that rank) every other
    query moves by 2e-5 .. 1e-4 through the inter-query attention.  A checked frame whose winners are not rank-identical must
    show the same winner SET with every moved rank on an oracle gap < 1e-4; the oracle then evaluates the frame once more
    with the winners in the HIP path's order (oracle.detect_frames(topk_override=...)) and everything is held to the strict
    tolerances again.
"""
import os
import time

import numpy as np
import pytest
import torch

from gomatching_amd.config import setup_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda"
FULL = os.environ.get("GOM_FULL_ORACLE_CLIP") == "1"


def _tc():
    from gomatching_amd.predictor import new_time_cost
    return new_time_cost()


def _prepare(cfg, frames_rgb):
    """Harness preparation per frame (each frame keeps its own source size)."""
    from gomatching_amd.predictor import GoMBatchPredictor
    inputs, sizes = [], []
    for f in frames_rgb:
        x, hw = GoMBatchPredictor(cfg, None).prepare([f[:, :, ::-1]])
        inputs.append(x[0])
        sizes.append(hw)
    return inputs, sizes


def _calibrate(cfg, image, frac, seed=2):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_fullsize_gpu import _calibrated_sd
    return _calibrated_sd(cfg, seed=seed, image=image, frac=frac)


def _px_tol(images):
    return max(1e-3, 8.0 * float(np.spacing(np.float32(max(max(im.shape[-2:]) for im in images)))))


def _rank_swaps(model, step_inputs, b, taps_o, nq):
    """Ranks at which the HIP path's proposal winners of frame `b` of a detector step differ from the oracle's, after checking
    that they are the same SET and that every moved rank sits on an oracle logit gap below 1e-4.  The WHOLE step is run again
    (eagerly, with taps): a near-tie can fall either way between two correct evaluations, e.g. a batch of 8 and a batch of 1
    (the stride-2 input_proj convolution slices K by the number of tiles), so the winners are read from the same batch shape
    the detections came from."""
    taps = {}
    x, _ = model.preprocess_image(step_inputs)
    feats = model.backbone.forward(x)
    model.detection_transformer.forward([feats[k] for k in model.feature_names], taps=taps)
    got = taps["topk"].reshape(len(step_inputs), nq)[b].cpu().numpy()
    ref = taps_o["topk"].reshape(-1).numpy()
    assert sorted(got.tolist()) == sorted(ref.tolist()), "proposal winners differ as a SET"
    moved = np.nonzero(got != ref)[0]
    val = taps_o["enc_class"].reshape(-1).numpy()[ref]                  # the oracle's logits in rank order (descending)
    for i in moved:
        j = int(np.nonzero(ref == got[i])[0][0])
        assert abs(float(val[i]) - float(val[j])) < 1e-4, ("rank %d <-> %d swapped across a gap of %.2e" % (i, j, abs(float(val[i] - val[j]))))
    return moved, got


def _oracle_insts(res):
    """The HIP path's per-frame detections as oracle `Inst`s (embeddings + boxes + payload), for the oracle's tracker."""
    from oracle import gom_oracle as O
    out = []
    for r in res:
        out.append(O.Inst(tuple(r.image_size), reid_features=r.reid_features.detach().cpu().clone(),
                          pred_boxes=r.pred_boxes.tensor.detach().cpu().clone(), scores=r.scores.detach().cpu().clone(),
                          ctrl_points=r.ctrl_points.detach().cpu().clone().flatten(1), recs=r.recs.detach().cpu().clone(),
                          bd=r.bd.detach().cpu().clone(), pred_classes=r.pred_classes.detach().cpu().clone()))
    return out


def _same_detections(got, ref, px_tol):
    assert len(got) == len(ref), (len(got), len(ref))
    if len(ref) == 0:
        return 0.0
    assert torch.equal(got["recs"], ref["recs"])
    # scores: sigmoid of the mean point logit after 12 transformer layers (north_star's bound is 1e-3)
    assert float((got["scores"] - ref["scores"]).abs().max()) <= 2e-5
    worst = 0.0
    for k in ("bd", "ctrl_points", "pred_boxes"):
        d = float((got[k] - ref[k]).abs().max())
        worst = max(worst, d)
        assert d <= px_tol, (k, d)
    assert float((got["reid_features"] - ref["reid_features"]).abs().max()) <= 1e-4
    return worst


# The oracle's detector on a full-size frame is 10-20 s of CPU work; the rank-conditioned and the unconditioned test of a clip
# look at the same frames of the same clip under the same weights, so each frame is detected once per session and shared
# (deep copies: the oracle's tracker writes into its inputs).  The taps kept are the two the rank test reads.
_ORACLE_DETS = {}


def _oracle_detect(key, sd, ocfg, image, f):
    import copy
    from oracle import gom_oracle as O
    k = key + (f,)
    if k not in _ORACLE_DETS:
        taps = {}
        with torch.no_grad():
            det = O.detect_frames(sd, ocfg, [image], taps=taps)[0]
        _ORACLE_DETS[k] = (det, {"topk": taps["topk"].clone(), "enc_class": taps["enc_class"].clone()})
    det, taps = _ORACLE_DETS[k]
    return copy.deepcopy(det), taps


def _clip_vs_oracle(builtin, frames_rgb, frac, check_frames, log, full=FULL, opts=()):
    """Runs the clip through `GoMatching.batch_inference` + short-track removal + rescaling and through the oracle.
    `full`: `oracle.run_clip` over EVERY frame, the oracle detecting on its own in its own rank order (unconditioned)."""
    FULL = full
    from oracle import gom_oracle as O
    from gomatching_amd.modeling import GoMatching
    cfg = setup_cfg(builtin=builtin, opts=opts)
    cfg.MODEL.DEVICE = DEV
    ocfg = setup_cfg(builtin=builtin, opts=opts)
    ocfg.MODEL.DEVICE = "cpu"
    inputs, sizes = _prepare(cfg, frames_rgb)
    images = [x["image"] for x in inputs]
    px_tol = _px_tol(images)
    sd = _calibrate(cfg, images[0], frac)
    model = GoMatching(cfg, sd, device=DEV, frames_per_step=8)
    # the detections alone first (tracking drops the embeddings of frames that left the window, gom_lstmatcher.py:402-403):
    # they feed the oracle's tracker; `batch_inference` then detects the same frames again (same bits: test_determinism_gpu.py)
    model.begin_batch([], len(inputs))
    raw = _oracle_insts(model.detect_steps(inputs, _tc()))
    t0 = time.time()
    insts, id_count = model.batch_inference(inputs, 0, 0, [], _tc())
    torch.cuda.synchronize()
    log["gpu_s"] = time.time() - t0
    assert model.fallback_steps == 0
    for a, b in zip(raw, insts):
        assert torch.equal(a["scores"], b.scores.cpu()) and torch.equal(a["pred_boxes"], b.pred_boxes.tensor.cpu())
    raw_ids = [x.track_ids.cpu().clone() for x in insts]
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    import hashlib
    key = (builtin, tuple(opts), float(frac), len(frames_rgb), hashlib.sha1(np.ascontiguousarray(frames_rgb[0]).tobytes()).hexdigest())
    t0 = time.time()
    log["checked_frames"] = []
    if not FULL:
        with torch.no_grad():
            for f in check_frames:                                     # the CPU detector on a subset of the frames
                ref, taps_o = _oracle_detect(key, sd, ocfg, images[f], f)
                s0, s1 = [st for st in model._steps(inputs) if st[0] <= f < st[1]][0]
                moved, order = _rank_swaps(model, inputs[s0:s1], f - s0, taps_o, cfg.MODEL.TRANSFORMER.NUM_QUERIES)
                if len(moved):                                         # near-tied winners fell the other way: same order, again
                    ref = O.detect_frames(sd, ocfg, [images[f]], topk_override=order)[0]
                worst = _same_detections(raw[f], ref, px_tol)
                log["checked_frames"].append({"frame": f, "detections": len(ref), "ranks_moved": moved.tolist(), "max_abs_px": worst})
    kept = model._remove_short_track(list(insts)) if model.min_track_len > 0 else insts
    res = model.batch_postprocess(kept, sizes)
    log["detections"] = [len(x) for x in raw]
    log["kept"] = [len(r["instances"]) for r in res]
    log["tracks"] = int(id_count)
    with torch.no_grad():
        if FULL:
            per_frame = [_oracle_detect(key, sd, ocfg, im, f)[0] for f, im in enumerate(images)]
            o_res, o_count = O.run_clip(sd, ocfg, images, orig_hw=sizes, per_frame=per_frame)
        else:
            # ids: identical, or parting only at near-ties of the ORACLE's own traj matrix (gap < 1e-4) whose forcing reproduces
            # every later id (helpers.track_clip_tie_aware); the report goes into the log: forced decisions, the clip's margins
            from helpers import track_clip_tie_aware
            o_inst, o_count, rep = track_clip_tie_aware(sd, ocfg, raw, [r.tolist() for r in raw_ids])
            log["tracker_forced_near_ties"], log["tracker_margins"] = rep["forced"], rep["margins"]
            if ocfg.VIDEO_TEST.MIN_TRACK_LEN > 0:
                o_inst = O.remove_short_track(ocfg, o_inst)
            o_res = O.batch_postprocess(o_inst, sizes)
    log["oracle_s"] = time.time() - t0
    assert int(id_count) == int(o_count)
    mx = 0.0
    for f, (r, g) in enumerate(zip(o_res, res)):
        r, g = r["instances"], g["instances"]
        assert len(r) == len(g), ("frame", f, len(r), len(g))
        if len(r) == 0:
            continue
        assert g.track_ids.cpu().tolist() == r["track_ids"].tolist(), ("ids", f)
        assert torch.equal(g.recs.cpu(), r["recs"]), ("characters", f)
        d = max(float((g.bd.cpu() - r["bd"]).abs().max()), float((g.ctrl_points.cpu().flatten(1) - r["ctrl_points"]).abs().max()))
        mx = max(mx, d)
        # FULL: the oracle detected on its own, in its own rank order -- where near-tied proposal winners fell differently (above),
        # every query of the frame has moved by up to ~1e-4 of the image size: north_star's bound on normalised coordinates (1e-3)
        side = max(max(im.shape[-2:]) for im in images)
        assert d <= (1e-3 * side if FULL else px_tol), \
            ("points: frame %d moved %.3e px = %.2e of the %d-px image side; north_star's bound is 1e-3 on NORMALISED coordinates "
             "in the unconditioned run (a near-tied proposal rank swap moves every query of the frame by ~1e-4 of the image), "
             "%.1e px in the rank-conditioned one" % (f, d, d / side, side, px_tol))
        log.setdefault("max_norm", 0.0)
        log["max_norm"] = max(log["max_norm"], d / side)
    log["max_abs_px"] = mx
    log["px_tol"] = px_tol
    log["mode"] = "oracle.run_clip over every frame" if FULL else \
        "oracle detector on frames %s, oracle tracker over the HIP detections of all frames" % (list(check_frames),)
    model.close()
    return log


def test_dstext_clip_300_queries_vs_oracle():
    """Config #4 as a tracked clip: the matcher at n up to 300, SHA_FFN_CRSATTN, NMS 0.3, long-term windows of 6 frames."""
    from gomatching_amd.synth import make_clip
    frames = make_clip(8, 1080, 1920, clip_id=4, num_rects=14)
    log = _clip_vs_oracle("pp_dstext", frames, 0.3, check_frames=(0, 5), log={"config": "pp_dstext 8 x 1920x1080 -> 1280x2276, nq 300"})
    print("CLIP", log)
    assert max(log["detections"]) >= 20 and log["tracks"] > max(log["detections"])


def test_dstext_clip_unconditioned_oracle_on_every_frame():
    """VERDICT r4 "weak" 1: the same DSText clip with NOTHING of the oracle conditioned on the HIP path -- `oracle.run_clip`
    detects and tracks all 8 frames on its own (its own top-k order).  ids and characters identical; points within 1e-3 of the
    image side (normalised coordinates, north_star's unit): where two near-tied proposal winners fall the other way between
    the two fp32 evaluations every query of that frame moves by ~1e-4 of the image (measured 0.16 px = 7e-5 at 2276 px)."""
    from gomatching_amd.synth import make_clip
    frames = make_clip(8, 1080, 1920, clip_id=4, num_rects=14)
    log = _clip_vs_oracle("pp_dstext", frames, 0.3, check_frames=(), log={"config": "pp_dstext, unconditioned"}, full=True)
    print("CLIP", log)
    assert log["max_norm"] <= 1e-3 and log["tracks"] > max(log["detections"])


def test_bovtext_clip_unconditioned_oracle_on_every_frame():
    """As above for config #5 (mixed-resolution sources, voc 5462)."""
    from gomatching_amd.synth import make_clip
    frames = make_clip(3, 720, 1280, clip_id=6, num_rects=10) + make_clip(3, 1080, 1920, clip_id=6, num_rects=10) + \
        make_clip(2, 1280, 720, clip_id=6, num_rects=10)
    log = _clip_vs_oracle("bovtext", frames, 0.7, check_frames=(), log={"config": "bovtext, unconditioned"}, full=True)
    print("CLIP", log)
    assert log["max_norm"] <= 1e-3


def test_dstext_tracker_stress_every_query_a_detection():
    """The tracker-stress variant of SURVEY.md 8-d at config #4: the class bias lets EVERY query through the threshold, so
    frames carry up to 300 detections before NMS and the long-term windows approach 6 x 300 rows."""
    from gomatching_amd.synth import make_clip
    frames = make_clip(8, 1080, 1920, clip_id=5, num_rects=14)
    log = _clip_vs_oracle("pp_dstext", frames, 1.0, check_frames=(), log={"config": "pp_dstext stress (every query passes)"})
    print("CLIP", log)
    assert max(log["detections"]) >= 60                              # (of 300 queries through the threshold, NMS 0.3 leaves ~75)


def test_dstext_lstmatcher_300_queries_stress_vs_oracle_tracker():
    """VERDICT r4 "missing" 4: `GoMatching_DSText.yaml` = the DSText geometry with the LSTMatcher head (separate short / long
    matcher transformers, 128-dim heads) at 300 queries -- long-term windows of up to 6 x ~75 rows after NMS go through
    `mha_core`'s 128-dim-head path.  Stress bias (every query passes the threshold); ids vs the oracle's tracker on all frames."""
    from gomatching_amd.synth import make_clip
    frames = make_clip(8, 1080, 1920, clip_id=5, num_rects=14)
    log = _clip_vs_oracle("pp_dstext", frames, 1.0, check_frames=(), log={"config": "GoMatching_DSText (LSTMatcher, nq 300), stress"},
                          opts=("MODEL.ROI_HEADS.NAME", "LSTMatcher", "MODEL.ASSO_HEAD.ASSO_THRESH_TEST", "0.5"))
    print("CLIP", log)
    assert max(log["detections"]) >= 60


def test_bovtext_mixed_resolution_clip_vs_oracle():
    """Config #5: bilingual head (voc 5462), one `batch_inference` call over 1280x720, 1920x1080 and 720x1280 sources."""
    from gomatching_amd.synth import make_clip
    a = make_clip(3, 720, 1280, clip_id=6, num_rects=10)
    b = make_clip(3, 1080, 1920, clip_id=6, num_rects=10)               # same scene generator key: rectangles persist in
    c = make_clip(2, 1280, 720, clip_id=6, num_rects=10)                # relative position across the size changes
    frames = a + b + c
    log = _clip_vs_oracle("bovtext", frames, 0.7, check_frames=(1, 4, 7), log={"config": "bovtext voc 5462, sources 3 x 1280x720 + 3 x 1920x1080 + 2 x 720x1280"})
    print("CLIP", log)
    assert min(log["detections"]) >= 3


def test_64_frame_clip_as_eight_shards_on_one_gpu():
    """Config #3's logic without the 8-GPU node: the eight ranks' work done one after the other on ONE GPU -- eight 8-frame
    shards detected separately, each packed into the all-gather record (`pack_records`), the eight buffers concatenated in
    rank order (what `all_gather_into_tensor` returns), unpacked and tracked once (`exchange_and_track`'s second half).  Track
    ids must equal those of the single-process `batch_inference` over the 64 frames AND the oracle's tracker."""
    from oracle import gom_oracle as O
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.synth import make_clip
    from gomatching_amd.dist import pack_records, unpack_records
    cfg = setup_cfg(builtin="icdar15")
    cfg.MODEL.DEVICE = DEV
    ocfg = setup_cfg(builtin="icdar15")
    ocfg.MODEL.DEVICE = "cpu"
    frames = make_clip(64, 720, 1280, clip_id=3, num_rects=12)
    inputs, sizes = _prepare(cfg, frames)
    sd = _calibrate(cfg, inputs[0]["image"], 0.3)
    T = cfg.MODEL.TRANSFORMER
    model = GoMatching(cfg, sd, device=DEV, frames_per_step=8)
    single, count_single = model.batch_inference(inputs, 0, 0, [], _tc())
    ids_single = [x.track_ids.cpu().tolist() for x in single]
    model.begin_batch([], 64)
    raw = _oracle_insts(model.detect_steps(inputs, _tc()))
    recs = []
    for r in range(8):                                                 # "rank" r: its block of 8 frames
        model.begin_batch([], 8)
        dets = model.detect_steps(inputs[8 * r:8 * r + 8], _tc())
        recs.append(pack_records(dets, T.NUM_QUERIES, model.roi_heads.feature_dim, T.NUM_POINTS, model.device).clone())
    allrec = torch.cat(recs)                                           # rank order = frame order
    model.begin_batch([], 64)
    all_dets = unpack_records(allrec, single[0].image_size, model.roi_heads.feature_dim, T.NUM_POINTS)
    sharded, count_sharded = model.track_frames(all_dets, 0, 0, [], _tc())
    ids_sharded = [x.track_ids.cpu().tolist() for x in sharded]
    assert ids_sharded == ids_single and int(count_sharded) == int(count_single)
    with torch.no_grad():
        o_inst, o_count = O.track_clip(sd, ocfg, raw)
    assert [x["track_ids"].tolist() for x in o_inst] == ids_single and int(o_count) == int(count_single)
    kept = model._remove_short_track(sharded)
    o_kept = O.remove_short_track(ocfg, o_inst)
    assert [x.track_ids.cpu().tolist() for x in kept] == [x["track_ids"].tolist() for x in o_kept]
    n = [len(x) for x in ids_single]
    print("CLIP64 detections/frame %s tracks %d kept/frame %s" % (n, int(count_single), [len(x) for x in kept]))
    assert min(n) >= 5 and int(count_single) > max(n)
    model.close()
