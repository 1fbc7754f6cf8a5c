#!/usr/bin/env python
"""Benchmark of the GoMatching inference hot path on MI355X (contract: see the task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path -- the reference's timed window, GoMBatchPredictor.__call__ from
`batch_inference` through short-track removal and rescaling (text_track_visualizer.py:325-334) -- over
one synthetic clip whose frames are already resident in HBM.  Workload (BASELINE.json configs[1]):
1280x720 source frames -> harness resize to 1000x1778 (MIN_SIZE_TEST=1000), 8 frames per GPU,
GoMatching_ICDAR15 config (R-50, 100 queries, LSTMatcher, rescoring), random-init synthetic weights.
With N > 1 the clip has 8*N frames, block-sharded 8 per rank, one RCCL all-gather of per-frame
association records, tracker replicated (weak scaling).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAMES_PER_GPU = 8
SRC_HW = (720, 1280)
# MI355X_MICROARCH.md "Matrix cores": fp32-input MFMA 157.3 TFLOP/s dense; bf16 MFMA ~2500 TFLOP/s dense.  The
# bf16x6 kernel issues 6 bf16 MFMA passes per fp32-equivalent product, so its ceiling in ALGORITHMIC flops is 2500/6.
PEAKS = {"fp32": ("gemm_f32_kernel<128,128,64,64,0,0>", 157.3, 1),
         "bf16x6": ("gemm_bf16x6_kernel<128,128,0,0>", 2500.0 / 6.0, 6),
         "f16x3": ("gemm_f16x3_kernel<128,128,0,0,3>", 2500.0 / 3.0, 3)}
DTYPES = {"fp32": "f32 (exact fp32 MFMA)",
          "bf16x6": "f32 (bf16x6 split MFMA: 24-bit significand products, fp32 accumulate)",
          "f16x3": "f32 (f16x3 split MFMA: 22-bit significand products, fp32 accumulate; parity tests at fp32 tolerances)"}


def pmc_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 on
    gfx950 per MI355X_MICROARCH.md + WRITE_SIZE); bench.py cannot collect PMCs itself."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        return rec[kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def build_model(cfg, device):
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.weights import synth_state_dict
    sd = synth_state_dict(cfg, seed=0)
    return GoMatching(cfg, sd, device=device, frames_per_step=FRAMES_PER_GPU), sd


def calibrate(model, inputs, frac=0.3):
    """Random-init DeepSolo detects nothing (class bias = -log(99) in the reference); shift the class /
    rescoring biases once so that ~30 % of the queries pass the threshold.  Setup only."""
    from gomatching_amd.predictor import new_time_cost
    tc = new_time_cost()
    x, hw = model.preprocess_image(inputs[:1])
    feats = model.backbone.forward(x)
    out = model.detection_transformer.forward([feats[k] for k in model.feature_names])
    thr = model.test_score_threshold
    logit_thr = float(np.log(thr / (1 - thr)))
    T = model.cfg.MODEL.TRANSFORMER
    m = out["pred_logits"].view(T.NUM_QUERIES, T.NUM_POINTS).mean(1)
    shift = logit_thr - float(torch.quantile(m, max(0.0, 1 - frac))) + (1.0 if frac >= 1.0 else 0.0)
    model.detection_transformer.ctrl_class[1].add_(shift)
    re_shift = None
    if model.with_rescore:
        r = model.roi_heads.rescoring_head(out["query_features"]).view(T.NUM_QUERIES, T.NUM_POINTS).mean(1)
        re_shift = logit_thr - float(torch.quantile(r, max(0.0, 1 - frac * 0.6)))
        model.roi_heads._rescoring[1].add_(re_shift)
    return shift, re_shift


def cpu_baseline(cfg, sd, shift, re_shift, frame_chw, gpu_frame0):
    """The CPU oracle ("port") timed on the host cores, on a bounded sample: ONE 1000x1778 frame through
    detection + embedding + id initialisation (short-track removal is skipped: it would delete every track of a
    1-frame sample).  The same run doubles as a full-size parity check of frame 0 against the HIP path."""
    from oracle import gom_oracle as O
    cores = min(os.cpu_count() or 1, 32)                    # torch's CPU kernels stop scaling (and thrash) beyond this
    torch.set_num_threads(cores)
    sd = dict(sd)
    k = "detection_transformer.ctrl_point_class.0.bias"
    sd[k] = sd[k] + shift
    if re_shift is not None:
        sd["roi_heads.rescoring_head.bias"] = sd["roi_heads.rescoring_head.bias"] + re_shift
    t0 = time.time()
    with torch.no_grad():
        dets = O.detect_frames(sd, cfg, [frame_chw])
        O.track_clip(sd, cfg, dets)
    dt = time.time() - t0
    ref = dets[0]
    n = len(ref)
    parity = {"detections_cpu": n, "detections_gpu": len(gpu_frame0)}
    if n == len(gpu_frame0) and n > 0:
        parity["max_abs_score"] = float((gpu_frame0.scores.cpu() - ref["scores"]).abs().max())
        parity["max_abs_bd_px"] = float((gpu_frame0.bd.cpu() - ref["bd"]).abs().max())
        parity["max_abs_ctrl_px"] = float((gpu_frame0.ctrl_points.cpu() - ref["ctrl_points"]).abs().max())
        parity["recs_identical"] = bool(torch.equal(gpu_frame0.recs.cpu(), ref["recs"]))
        parity["max_abs_reid"] = float((gpu_frame0.reid_features.cpu() - ref["reid_features"]).abs().max())
    return {"value": 1.0 / dt, "unit": "frames/sec", "cores": cores, "kind": "port",
            "sample": "1 frame 1280x720->1000x1778 through oracle/gom_oracle.py detect_frames+track_clip "
                      "(%.1f s of CPU work on %d threads)" % (dt, cores),
            "full_size_parity_frame0": parity}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backbone", default="r50", choices=["r50", "swin", "vitae"],
                    help="r50 = BASELINE.json's workload; swin = side measurement of the Swin-T backbone (§8-f3) on the "
                         "same frames (not the BASELINE workload)")
    ap.add_argument("--detect-frac", type=float, default=0.3,
                    help="fraction of the queries the calibrated biases let through the score threshold (SURVEY.md §8-d: "
                         "0.3 for the BASELINE workload; 1.0 = the tracker-stress variant, every query a detection before NMS)")
    ap.add_argument("--fused-matcher", type=int, default=0,
                    help="diagnostic: long-term matches through the persistent one-kernel matcher with this many workgroups")
    ap.add_argument("--h2d", default=None, choices=["kernel", "dma", "sync"],
                    help="diagnostic: how the tracker uploads its per-match descriptors (GoMatching.h2d_mode)")
    ap.add_argument("--emulate-world", type=int, default=1,
                    help="N=1 diagnostic: run the replicated tracker over W copies of this GPU's records per step, i.e. "
                         "the tracker load of a W-GPU run, beside one GPU's detection (value still counts 8 frames/step)")
    ap.add_argument("--gemm", default="f16x3", choices=["f16x3", "bf16x6", "fp32"],
                    help="contraction back-end: two-plane fp16 split on the fp16 matrix cores (default), three-plane bf16 "
                         "split, or exact-fp32 MFMA")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    import torch.distributed as dist
    # GOM_BENCH_BACKEND=gloo is a dry-run aid only: several ranks share the one GPU of a test box and exchange through host
    # memory, which exercises the whole N>1 code path except RCCL itself (the driver's multi-GPU runs use the default)
    backend = os.environ.get("GOM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from gomatching_amd import ops
    from gomatching_amd.config import setup_cfg
    from gomatching_amd.dist import sharded_batch_inference
    from gomatching_amd.predictor import GoMBatchPredictor, new_time_cost
    from gomatching_amd.synth import make_clip

    ops.GEMM_MODE = args.gemm
    cfg = setup_cfg(builtin="icdar15")
    cfg.MODEL.DEVICE = "cuda"
    src_hw = SRC_HW
    if args.backbone == "swin":
        cfg.MODEL.BACKBONE.NAME = "build_swin_backbone"        # same frames and resize as the R-50 workload
    if args.backbone == "vitae":
        cfg.MODEL.BACKBONE.NAME = "build_vitaev2_backbone"
        src_hw = (1024, 1792)                                  # ViTAE needs multiples of 32 (the reference asserts): frames
        cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = 1024, 2000     # arrive at network size, the resize is a no-op
    model, sd = build_model(cfg, device)
    if args.fused_matcher:
        from gomatching_amd import lib as _lib
        _lib.load().gom_tracker_set_fused(1)
        _lib.load().gom_match_fused_set_grid(args.fused_matcher)
        ops.FUSED_MATCHER = True
    if args.h2d:
        model.h2d_mode = args.h2d
    predictor = GoMBatchPredictor(cfg, model)

    # this rank's block of the clip: frames [rank*8, rank*8+8) of a world*8-frame synthetic video
    clip = make_clip(FRAMES_PER_GPU * world, src_hw[0], src_hw[1], clip_id=0, num_rects=12)
    mine = [f[:, :, ::-1] for f in clip[rank * FRAMES_PER_GPU:(rank + 1) * FRAMES_PER_GPU]]   # harness takes BGR
    inputs, hw = predictor.prepare(mine)                       # host resize etc.: outside the timed window
    inputs = [dict(x, image=x["image"].to(device)) for x in inputs]      # resident in HBM before timing starts
    net_hw = tuple(inputs[0]["image"].shape[-2:])
    # every rank calibrates on the SAME frame (frame 0 of the clip) so that all ranks hold identical weights
    cal_inputs, _ = predictor.prepare([clip[0][:, :, ::-1]])
    cal_inputs = [dict(x, image=x["image"].to(device)) for x in cal_inputs]
    shift, re_shift = calibrate(model, cal_inputs, frac=args.detect_frac)

    from gomatching_amd.dist import exchange_and_track
    from gomatching_amd.predictor import ClipPipeline

    def finish(h):
        """Tracker half of a step (runs on the tracker stream, overlapping the next step's detection)."""
        model.begin_batch([], FRAMES_PER_GPU * world * args.emulate_world)
        dets = model.detect_finish(h, tc)
        if args.emulate_world > 1 and world == 1:
            from gomatching_amd.dist import pack_records, unpack_records
            T = cfg.MODEL.TRANSFORMER
            rec = pack_records(dets, T.NUM_QUERIES, model.roi_heads.feature_dim, T.NUM_POINTS, device)
            dets = unpack_records(torch.cat([rec] * args.emulate_world), dets[0].image_size, model.roi_heads.feature_dim,
                                  T.NUM_POINTS)
            insts, id_count = model.track_frames(dets, 0, 0, [], tc)
        elif world > 1:
            insts, id_count = exchange_and_track(model, dets, 0, 0, [], tc)
        else:
            insts, id_count = model.track_frames(dets, 0, 0, [], tc)
        if model.min_track_len > 0:
            insts = model._remove_short_track(insts)
        return model.batch_postprocess(insts, [hw] * len(insts)), id_count

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pipe = ClipPipeline(model, finish)
    tc = new_time_cost()
    for _ in range(2):                                         # setup, not warm-up: eager pass (per-resolution caches)
        pipe.push(inputs, tc)                                  # + hipGraph capture of the detector
    pipe.flush()
    for _ in range(args.warmup):
        pipe.push(inputs, tc)
    pipe.flush()
    tc = new_time_cost()
    barrier()
    t0 = time.time()
    done = 0
    for _ in range(args.steps):                                # K steps: detector(i+1) overlaps tracker(i)
        r = pipe.push(inputs, tc)
        if r is not None:
            res, id_count = r
            done += 1
    res, id_count = pipe.flush()
    done += 1
    barrier()
    elapsed = time.time() - t0
    assert done == args.steps
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    # Roofline leg.  The timed steps replay the detector as a hipGraph, which hides individual launches from HIP
    # events; the dominant kernel is therefore bracketed with events in PROFILE_STEPS eager steps of the same
    # workload right after the timed region (same process, same buffers, same stream).  profiles/ holds the rocprofv3
    # per-kernel average over the timed (graph) steps of this command, which agrees.
    graphed = bool(model.use_graphs and model._graphs)
    model.use_graphs = False
    prof = []
    ops.set_gemm_profile(prof)                                 # HIP events around the dominant kernel's launches
    tcp = new_time_cost()
    PROFILE_STEPS = 2
    for _ in range(PROFILE_STEPS):
        pipe.push(inputs, tcp)
    pipe.flush()
    barrier()
    ops.set_gemm_profile(None)
    model.use_graphs = graphed

    total_frames = FRAMES_PER_GPU * world * args.steps
    fps = total_frames / elapsed
    dur_ms = sum(p[0].elapsed_time(p[1]) for p in prof)
    flops = sum(p[2] for p in prof)
    alg_bytes = sum(p[3] for p in prof)
    achieved = flops / (dur_ms * 1e-3) / 1e12 if dur_ms > 0 else 0.0
    line = {
        "metric": "frames/sec (whole node), 1280x720 clip, 100 queries/frame",
        "value": fps, "unit": "frames/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPES[args.gemm], "data": "synthetic",
        "config": {"workload": ("configs[1]: 1280x720 ICDAR15-video clip -> %dx%d net input, %d frames/GPU, "
                                "100 queries, GoMatching_ICDAR15 (R-50 + DeepSolo + LSTMatcher, rescoring), "
                                "random-init synthetic weights" % (net_hw[0], net_hw[1], FRAMES_PER_GPU))
                   if args.backbone == "r50" else
                   ("NOT the BASELINE workload: %s backbone side measurement, %dx%d net input, %d frames/GPU, 100 "
                    "queries, DeepSolo + LSTMatcher, random-init synthetic weights"
                    % ({"swin": "Swin-T", "vitae": "ViTAEv2-S"}[args.backbone], net_hw[0], net_hw[1], FRAMES_PER_GPU)),
                   "frames_per_step": FRAMES_PER_GPU * world, "emulated_world": args.emulate_world, "pipelining": "detector(step i+1) overlaps tracker(step i)", "detector_hipgraph": graphed,
                   "parallelism": "frame-sharded dp%d + 1 all-gather/step"
                   % world if world > 1 else "single GPU",
                   "detect_frac": args.detect_frac,
                   "detections_per_frame": [len(r["instances"]) for r in res[:FRAMES_PER_GPU]],
                   "tracks": int(id_count)},
        "roofline": {"bound": "mfma", "kernel": PEAKS[args.gemm][0], "achieved": achieved,
                     "peak": PEAKS[args.gemm][1], "unit": "TFLOP/s", "frac": achieved / PEAKS[args.gemm][1],
                     "traffic": pmc_traffic(PEAKS[args.gemm][0]), "mfma_passes_per_product": PEAKS[args.gemm][2],
                     "peak_note": {"fp32": "dense fp32-input MFMA peak",
                                   "bf16x6": "algorithmic fp32-equivalent FLOP/s; dense bf16 MFMA peak 2500 / 6 passes",
                                   "f16x3": "algorithmic fp32-equivalent FLOP/s; dense fp16 MFMA peak 2500 / 3 passes"}[args.gemm], "launches_per_step": len(prof) // PROFILE_STEPS,
                     "avg_launch_us": dur_ms * 1e3 / max(len(prof), 1),
                     "flops_per_launch_avg": flops / max(len(prof), 1),
                     "algorithmic_bytes_per_launch_avg": alg_bytes / max(len(prof), 1),
                     "share_of_step_time": (dur_ms / PROFILE_STEPS) / (elapsed / args.steps * 1e3),
                     "measured_in": "%d eager steps after the timed region (timed steps are hipGraph replays)" % PROFILE_STEPS
                     if graphed else "%d eager steps after the timed region" % PROFILE_STEPS},
        "stage_ms_per_step": {k: v / args.steps * 1e3 for k, v in tc.items() if isinstance(v, float) and v > 0},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.backbone == "r50":
        cpu_cfg = setup_cfg(builtin="icdar15")
        cpu_cfg.MODEL.DEVICE = "cpu"
        gpu0 = model.inference(inputs[:1], new_time_cost())[0]
        line["cpu_baseline"] = cpu_baseline(cpu_cfg, sd, shift, re_shift, inputs[0]["image"].cpu(), gpu0)
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
