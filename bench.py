#!/usr/bin/env python
"""Benchmark of the GoMatching inference hot path on MI355X (contract: see the task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path -- the reference's timed window, GoMBatchPredictor.__call__ from
`batch_inference` through short-track removal and rescaling (text_track_visualizer.py:325-334) -- over
one synthetic clip.  `value` (round 6 on: the contract's definition) = the K steps with the resized fp32 frames already
resident in HBM when the timed region starts; `value_pcie_inclusive` = the same K steps with the frames in (pinned) host
memory and their H2D copy inside the window, on an upload stream under the previous step's detector -- the reference's
window as it stands, which rounds 1-5 reported as `value`.  Workload (BASELINE.json configs[1]):
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


def kernel_source_hash():
    """sha1 over the GEMM kernel sources: profiles/pmc_traffic.json records the hash of the build its counters were taken
    on (tools/pmc_traffic.py), and `roofline.traffic` is only printed while that build is still the current one."""
    import hashlib
    h = hashlib.sha1()
    for name in ("gemm_f16x3.hip", "gemm_bf16x6.hip", "gemm_conv.hip", "gemm_k256.hip", "ffn_fused.hip", "proj_ln.hip", "msda.hip",
                 "dec_attn.hip", "dec_attn2.hip", "dec_tail.hip", "dec_tail2.hip", "bneck_fused.hip", "bneck2.hip", "conv3x3_patch.hip", "common.h"):
        with open(os.path.join(ROOT, "gomatching_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel):
    """HBM bytes per launch of the dominant kernel's GEMM-API launches from the committed rocprofv3 PMC passes (FETCH_SIZE
    x2 on gfx950 per MI355X_MICROARCH.md + WRITE_SIZE); bench.py cannot collect PMCs itself.  None when the counters
    belong to another build of the kernel."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        if rec.get("_meta", {}).get("kernel_source_hash") != kernel_source_hash():
            return None
        return rec[kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def ffn_traffic(calls):
    """HBM bytes per fused-FFN CALL from the PMC passes: the 128-row-tile launch + (long launches only) its half-height tail launch,
    weighted by how many of the step's calls have one."""
    from gomatching_amd import ops
    if not (getattr(ops, "DEC_TAIL", True) and getattr(ops, "FUSED_FFN", True)):
        return None                                          # an ablation run (GOM_DEC_TAIL=0): the decoder's short FFN launches are in the
                                                             # population and the per-call counters of the committed passes do not apply
    main = pmc_traffic("ffn_fused_kernel<false,2>")
    tail = pmc_traffic("ffn_fused_kernel<false,1>")
    if main is None:
        return None
    # (round 5: the decoder's FFN blocks run inside the tail launch, csrc/dec_tail.hip -- every remaining call is an encoder call =
    # the 128-row-tile launch + its half-height tail launch)
    return main + (tail if tail is not None else 0.0)


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
    model.add_class_bias(shift)
    re_shift = None
    if model.with_rescore:
        r = model.roi_heads.rescoring_head(out["query_features"]).view(T.NUM_QUERIES, T.NUM_POINTS).mean(1)
        re_shift = logit_thr - float(torch.quantile(r, max(0.0, 1 - frac * 0.6)))
        model.roi_heads._rescoring[1].add_(re_shift)
    return shift, re_shift


def cpu_baseline(cfg, sd, shift, re_shift, frames_chw, orig_hw, gpu_res, gpu_id_count):
    """The CPU oracle ("port") timed on the host cores over the WHOLE clip of one step (8 frames of 1000x1778 through
    detection, embedding, tracker, short-track removal and rescaling: oracle/gom_oracle.py run_clip), and the same run as
    the full-size parity check of the HIP path: per frame the detections, characters and TRACK IDS must be identical,
    points within 1e-3 px (north_star)."""
    from oracle import gom_oracle as O
    cores = min(os.cpu_count() or 1, 32)                    # torch's CPU kernels stop scaling (and thrash) beyond this
    torch.set_num_threads(cores)
    sd = dict(sd)
    k = "detection_transformer.ctrl_point_class.0.bias"
    sd[k] = sd[k] + shift
    if re_shift is not None:
        sd["roi_heads.rescoring_head.bias"] = sd["roi_heads.rescoring_head.bias"] + re_shift
    # MatchLog: how close the clip's discrete tracker decisions sit to a flip -- smallest gap between the chosen assignment's total
    # score and the best assignment with another outcome, smallest |score - threshold| over the chosen pairs
    # (gom_lstmatcher.py:434-452, 521-554), in the oracle's traj.  Its ~n extra assignment solves per match are not the reference's
    # work: the log times itself and that time is taken out of the baseline's
    mlog = O.MatchLog()
    t0 = time.time()
    ref, ref_idc = O.run_clip(sd, cfg, frames_chw, orig_hw=orig_hw, log=mlog)
    dt = time.time() - t0 - mlog.seconds
    parity = {"frames": len(ref), "id_count_cpu": int(ref_idc), "id_count_gpu": int(gpu_id_count),
              "detections_cpu": [len(r["instances"]) for r in ref],
              "detections_gpu": [len(r["instances"]) for r in gpu_res]}
    same_n = parity["detections_cpu"] == parity["detections_gpu"]
    ids_same = recs_same = same_n
    mx = {"score": 0.0, "bd_px": 0.0, "ctrl_px": 0.0}
    if same_n:
        for r, g in zip(ref, gpu_res):
            r, g = r["instances"], g["instances"]
            if len(r) == 0:
                continue
            ids_same = ids_same and bool(torch.equal(g.track_ids.cpu(), r["track_ids"]))
            recs_same = recs_same and bool(torch.equal(g.recs.cpu(), r["recs"]))
            mx["score"] = max(mx["score"], float((g.scores.cpu() - r["scores"]).abs().max()))
            mx["bd_px"] = max(mx["bd_px"], float((g.bd.cpu() - r["bd"]).abs().max()))
            mx["ctrl_px"] = max(mx["ctrl_px"], float((g.ctrl_points.cpu() - r["ctrl_points"]).abs().max()))
    parity.update({"track_ids_identical": ids_same, "recs_identical": recs_same, "max_abs_score": mx["score"],
                   "max_abs_bd_px": mx["bd_px"], "max_abs_ctrl_px": mx["ctrl_px"], "tracker_margins": mlog.summary()})
    n = len(frames_chw)
    return {"value": n / dt, "unit": "frames/sec", "cores": cores, "kind": "port",
            "sample": "the whole %d-frame clip of one step (1280x720 -> %dx%d) through oracle/gom_oracle.py run_clip: detector, "
                      "embedding, tracker, short-track removal, rescaling (%.1f s of CPU work on %d threads)"
                      % (n, frames_chw[0].shape[-2], frames_chw[0].shape[-1], dt, cores),
            "full_size_parity_clip": parity}


LEGS = {"dstext": ("pp_dstext", [(1080, 1920)] * 8,
                   "configs[3]: GoMatching_PP_DSText, 8 frames 1920x1080 -> 1280x2276, 300 queries, SHA_FFN_CRSATTN, no rescoring, NMS 0.3"),
        "bovtext": ("bovtext", [(720, 1280)] * 3 + [(1080, 1920)] * 3 + [(1280, 720)] * 2,
                    "configs[4]: GoMatching_BOVText, voc 5462 (bilingual head), ONE clip mixing 3 x 1280x720 + 3 x 1920x1080 + 2 x 720x1280 "
                    "sources (-> 1000x1778 / 1000x1778 / 1778x1000), 100 queries, LSTMatcher")}


def config_leg(leg, device, gemm, detect_frac, steps=6, warmup=2):
    """Secondary figure for BASELINE.json configs[3] / configs[4] on ONE GPU: the reference's timed window
    (text_track_visualizer.py:325-334: `batch_inference` of the clip + short-track removal + rescaling, frames in pinned host
    memory, H2D inside) over an 8-frame synthetic clip, `steps` times back to back after `warmup`.  Each clip is one
    `batch_inference` call (mixed sizes are split into per-size detector steps inside it, as the reference loops frames), so
    unlike the headline there is no overlap ACROSS clips.  Also returns the per-stage split of one eager, synchronised clip."""
    from gomatching_amd import ops
    from gomatching_amd.config import setup_cfg
    from gomatching_amd.predictor import GoMBatchPredictor, new_time_cost
    from gomatching_amd.synth import make_clip
    builtin, sizes, what = LEGS[leg]
    cfg = setup_cfg(builtin=builtin)
    cfg.MODEL.DEVICE = "cuda"
    ops.GEMM_MODE = gemm
    frames, t_of = [], {}
    for hw in sizes:                                             # one scene per source size, consecutive time indices
        t = t_of.get(hw, 0)
        t_of[hw] = t + 1
        frames.append((hw, t))
    clips = {hw: make_clip(n, hw[0], hw[1], clip_id=40 + len(leg), num_rects=12) for hw, n in t_of.items()}
    pred = GoMBatchPredictor(cfg, None)
    inputs, src = [], []
    for hw, t in frames:
        x, shw = pred.prepare([clips[hw][t][:, :, ::-1]])
        inputs.append(dict(x[0], image=x[0]["image"].pin_memory()))
        src.append(shw)
    model, sd = build_model(cfg, device)
    calibrate(model, [dict(inputs[0], image=inputs[0]["image"].to(device))], frac=detect_frac)

    def clip(tc):
        insts, id_count = model.batch_inference(inputs, 0, 0, [], tc)
        if model.min_track_len > 0:
            insts = model._remove_short_track(insts)
        return model.batch_postprocess(insts, src), id_count

    for _ in range(2 + warmup):                                  # eager pass + hipGraph capture per step shape, then warm-up
        clip(new_time_cost())
    torch.cuda.synchronize()
    tc = new_time_cost()
    t0 = time.time()
    for _ in range(steps):
        res, id_count = clip(tc)
    torch.cuda.synchronize()
    dt = time.time() - t0
    graphed = bool(model.use_graphs and any(isinstance(v, dict) for v in model._graphs.values()))
    # per-stage split: eager clips with a sync after every stage.  The FIRST eager clip after graph replays pays cold allocations
    # (the graph's private pool does not serve eager launches: 254 ms against 59 ms per replayed clip on the driver's box in round 4),
    # so it is run and dropped; the second one is reported
    use_graphs = model.use_graphs
    model.use_graphs = False
    for _ in range(2):
        tcs = new_time_cost(sync=True)
        t1 = time.time()
        clip(tcs)
        torch.cuda.synchronize()
        eager_ms = (time.time() - t1) * 1e3
    model.use_graphs = use_graphs
    out = {"value": len(inputs) * steps / dt, "unit": "frames/sec", "ms_per_clip": dt / steps * 1e3, "steps": steps, "warmup": warmup,
           "workload": what, "net_input_hw": sorted({tuple(x["image"].shape[-2:]) for x in inputs}),
           "detector_steps_per_clip": [b - a for a, b in model._steps(inputs)],
           "detector_hipgraph": graphed, "fallback_steps": int(model.fallback_steps),
           "fused_inter_attention": all(L["inter_block"] is not None for L in model.detection_transformer.dec),
           "detections_per_frame": [len(r["instances"]) for r in res], "tracks": int(id_count),
           "stage_ms_per_clip_eager_synced": dict({k: v * 1e3 for k, v in tcs.items() if isinstance(v, float) and v > 0},
                                                  whole_clip=eager_ms),
           "note": "secondary figure (headline = configs[1]); one batch_inference call per clip, clips back to back without "
                   "cross-clip overlap; parity of this configuration: tests/test_clips_fullsize_gpu.py"}
    assert model.fallback_steps == 0 or gemm != "f16x3"
    model.close()
    del model
    torch.cuda.empty_cache()
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` outside torchrun: start N fresh child processes of this script, one per GPU (RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment; the reference starts its own workers the same way,
    train_net.py:198-209 -> detectron2 `launch`), relay their output -- rank 0 prints the one JSON line -- and return non-zero
    when any rank fails.  The parent never touches the GPU (device_count() does not initialise HIP on this image), and
    nothing is exec'ed from a process that has."""
    import socket
    import subprocess
    backend = os.environ.get("GOM_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < n:
        print("bench.py: --gpus %d but only %d GPU(s) visible (RCCL needs one device per rank; GOM_BENCH_BACKEND=gloo lets "
              "the ranks share a device for a dry run)" % (n, have), file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:                                               # a dead rank would leave the others in a collective for ever
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print("bench.py: rank %d exited with %d; stopping the other ranks" % (procs.index(p), code), file=sys.stderr)
                for q in alive:
                    q.terminate()
        time.sleep(0.2)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-backends", action="store_true",
                    help="skip the short secondary measurements of the other two contraction back-ends (N=1 only)")
    ap.add_argument("--inputs", default="hbm", choices=["host", "hbm"],
                    help="where the frames are when the timed region of `value` starts: hbm (default: the contract's definition) or "
                         "pinned host memory with the H2D inside the window (the reference's window; what rounds 1-5 printed as "
                         "`value`); the other placement is always timed too (value_hbm_resident / value_pcie_inclusive)")
    ap.add_argument("--backbone", default="r50", choices=["r50", "swin", "vitae"],
                    help="r50 = BASELINE.json's workload; swin = side measurement of the Swin-T backbone (§8-f3) on the "
                         "same frames (not the BASELINE workload)")
    ap.add_argument("--detect-frac", type=float, default=0.3,
                    help="fraction of the queries the calibrated biases let through the score threshold (SURVEY.md §8-d: "
                         "0.3 for the BASELINE workload; 1.0 = the tracker-stress variant, every query a detection before NMS)")
    ap.add_argument("--h2d", default=None, choices=["kernel", "dma", "sync"],
                    help="diagnostic: how the tracker uploads its per-match descriptors (GoMatching.h2d_mode)")
    ap.add_argument("--tracker-cus", type=int, default=-1,
                    help="compute units reserved for the tracker's per-frame recurrence (CU-masked streams); default 0 = no lane.  "
                         "Rounds 2-5 reserved 32 from 8 GPUs' frames per tracker on (48.6 -> 42.2 ms per step in round 2); with "
                         "round 6's kernels the lane LOSES at 8 GPUs' load: 32.3-32.6 ms against 30.5-30.7 without it (three "
                         "alternating pairs, one box, N = 1: 28.05)")
    ap.add_argument("--emulate-world", type=int, default=1,
                    help="N=1 diagnostic: run the replicated tracker over W copies of this GPU's records per step, i.e. "
                         "the tracker load of a W-GPU run, beside one GPU's detection (value still counts 8 frames/step)")
    ap.add_argument("--replicate-short-term", action="store_true",
                    help="multi-GPU / --emulate-world: every rank scores EVERY frame pair of the clip (rounds 1-4) instead of its own "
                         "8 pairs + a second all-gather of the score blocks (dist.exchange_and_track, default)")
    ap.add_argument("--gemm", default="f16x3", choices=["f16x3", "bf16x6", "fp32"],
                    help="contraction back-end: two-plane fp16 split on the fp16 matrix cores (default), three-plane bf16 "
                         "split, or exact-fp32 MFMA")
    ap.add_argument("--config", default="all", choices=["ic15", "dstext", "bovtext", "all"],
                    help="the headline, BASELINE.json configs[1], is always measured.  ic15 = the headline ONLY; dstext / bovtext ADD the "
                         "secondary leg of configs[3] (GoMatching_PP_DSText: 1920x1080 -> 1280x2276, 300 queries) / configs[4] "
                         "(GoMatching_BOVText: voc 5462, mixed-resolution clip) as `value_dstext` / `value_bovtext`; all (default) = both")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the secondary configs[3] / configs[4] legs of the default run")
    ap.add_argument("--frames-per-gpu", type=int, default=FRAMES_PER_GPU,
                    help="diagnostic: frames of the clip each rank owns per step (BASELINE.json: 8; the self-launch test compares "
                         "N=2 x 8 with N=1 x 16, the same 16-frame clip)")
    args = ap.parse_args()
    globals()["FRAMES_PER_GPU"] = args.frames_per_gpu

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))              # plain `python bench.py --gpus N`: spawn the ranks ourselves
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
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
    from gomatching_amd.predictor import GoMBatchPredictor, new_time_cost
    from gomatching_amd.synth import make_clip
    from gomatching_amd.dist import exchange_and_track
    from gomatching_amd.predictor import ClipPipeline

    cfg = setup_cfg(builtin="icdar15")
    cfg.MODEL.DEVICE = "cuda"
    src_hw = SRC_HW
    if args.backbone == "swin":
        cfg.MODEL.BACKBONE.NAME = "build_swin_backbone"        # same frames and resize as the R-50 workload
    if args.backbone == "vitae":
        cfg.MODEL.BACKBONE.NAME = "build_vitaev2_backbone"
        src_hw = (1024, 1792)                                  # ViTAE needs multiples of 32 (the reference asserts): frames
        cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = 1024, 2000     # arrive at network size, the resize is a no-op

    # this rank's block of the clip: frames [rank*8, rank*8+8) of a world*8-frame synthetic video
    clip = make_clip(FRAMES_PER_GPU * world, src_hw[0], src_hw[1], clip_id=0, num_rects=12)
    mine = [f[:, :, ::-1] for f in clip[rank * FRAMES_PER_GPU:(rank + 1) * FRAMES_PER_GPU]]   # harness takes BGR
    host_inputs, hw = GoMBatchPredictor(cfg, None).prepare(mine)      # host resize etc.: outside the timed window
    host_inputs = [dict(x, image=x["image"].pin_memory()) for x in host_inputs]
    hbm_inputs = [dict(x, image=x["image"].to(device)) for x in host_inputs]
    net_hw = tuple(host_inputs[0]["image"].shape[-2:])
    cal_inputs, _ = GoMBatchPredictor(cfg, None).prepare([clip[0][:, :, ::-1]])
    cal_inputs = [dict(x, image=x["image"].to(device)) for x in cal_inputs]
    shifts = {}

    def make(mode):
        """Model + pipeline of one contraction back-end; every rank (and every back-end) calibrates on frame 0 of the
        clip, the default back-end's shifts being reused so that all of them hold identical weights."""
        ops.GEMM_MODE = mode
        model, sd = build_model(cfg, device)
        if args.h2d:
            model.h2d_mode = args.h2d
        cus = args.tracker_cus if args.tracker_cus >= 0 else 0
        if cus > 0 and device.type == "cuda":
            try:
                model.reserve_tracker_cus(cus)
            except Exception as e:                               # a scheduling aid only: never let it cost the run
                print("bench: CU reservation for the tracker unavailable (%s: %s); continuing without" % (type(e).__name__, e),
                      file=sys.stderr, flush=True)
                model.reserve_tracker_cus(0)
        if not shifts:
            shifts["s"], shifts["r"] = calibrate(model, cal_inputs, frac=args.detect_frac)
        else:
            model.add_class_bias(shifts["s"])
            if shifts["r"] is not None:
                model.roi_heads._rescoring[1].add_(shifts["r"])
        tc_box = [new_time_cost()]
        last_rec = [None]
        st_cache = [None]
        model._bench_last_rec = last_rec

        def finish(h):
            """Tracker half of a step (runs on the tracker stream, overlapping the next step's detection)."""
            tc = tc_box[0]
            w0 = time.time()
            model.begin_batch([], FRAMES_PER_GPU * world * args.emulate_world)
            h["event"].synchronize()
            w1 = time.time()
            dets = model.detect_finish(h, tc)
            w2 = time.time()
            if args.emulate_world > 1 and world == 1:
                from gomatching_amd.dist import pack_records, unpack_records, pack_short_term, unpack_short_term
                T = cfg.MODEL.TRANSFORMER
                rec = pack_records(dets, T.NUM_QUERIES, model.roi_heads.feature_dim, T.NUM_POINTS, device)
                last_rec[0] = (rec, dets[0].image_size)
                dets = unpack_records(torch.cat([rec] * args.emulate_world), dets[0].image_size,
                                      model.roi_heads.feature_dim, T.NUM_POINTS)
                w3 = time.time()
                st = None
                if not args.replicate_short_term:
                    # dist.exchange_and_track's sharded short-term precompute, emulated: THIS rank scores its own 8 frame pairs;
                    # the other ranks' blocks (copies of the same records: computed once, outside the measurement) arrive as one
                    # buffer copy standing for the second all-gather, and are unpacked (one D2H) as on a real rank
                    model._home_features(list(dets))
                    mine = list(range(FRAMES_PER_GPU))
                    blocks = pack_short_term(model.precompute_short_term(list(dets), only=set(mine)), mine, T.NUM_QUERIES, device)
                    if st_cache[0] is None:
                        st_cache[0] = pack_short_term(model.precompute_short_term(list(dets)), list(range(len(dets))),
                                                      T.NUM_QUERIES, device)
                    allblk = st_cache[0].clone()
                    allblk[:FRAMES_PER_GPU] = blocks
                    st = unpack_short_term(allblk, list(range(len(dets))))
                insts, id_count = model.track_frames(dets, 0, 0, [], tc, st=st)
            elif world > 1:
                w3 = time.time()
                insts, id_count = exchange_and_track(model, dets, 0, 0, [], tc, shard_short_term=not args.replicate_short_term)
            else:
                w3 = time.time()
                insts, id_count = model.track_frames(dets, 0, 0, [], tc)
            w4 = time.time()
            if model.min_track_len > 0:
                insts = model._remove_short_track(insts)
            out = model.batch_postprocess(insts, [hw] * len(insts)), id_count
            w5 = time.time()
            for key, dt in (("finish_wait_detector", w1 - w0), ("finish_embed", w2 - w1), ("finish_exchange", w3 - w2),
                            ("finish_track", w4 - w3), ("finish_post", w5 - w4)):
                tc[key] = tc.get(key, 0.0) + dt
            return out

        return model, sd, ClipPipeline(model, finish), tc_box

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(pipe, tc_box, inputs, steps, warmup):
        """W untimed + exactly K timed steps (detector(i+1) overlaps tracker(i)); max over ranks."""
        for _ in range(warmup):
            pipe.push(inputs, tc_box[0])
        pipe.flush()
        tc_box[0] = new_time_cost()
        barrier()
        t0 = time.time()
        done = 0
        for _ in range(steps):
            r = pipe.push(inputs, tc_box[0])
            if r is not None:
                res, id_count = r
                done += 1
        res, id_count = pipe.flush()
        done += 1
        barrier()
        elapsed = time.time() - t0
        assert done == steps
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t[0])
        return elapsed, res, id_count

    def setup(pipe, tc_box, inputs):
        for _ in range(2):                                     # setup, not warm-up: eager pass (per-resolution caches)
            pipe.push(inputs, tc_box[0])                       # + hipGraph capture of the detector
        pipe.flush()

    model, sd, pipe, tc_box = make(args.gemm)
    primary = host_inputs if args.inputs == "host" else hbm_inputs
    setup(pipe, tc_box, primary)
    elapsed, res, id_count = timed(pipe, tc_box, primary, args.steps, args.warmup)
    tc = tc_box[0]
    # the other placement of the frames, same K steps, as a secondary figure
    elapsed_other, _, _ = timed(pipe, tc_box, hbm_inputs if args.inputs == "host" else host_inputs, args.steps, 1)
    elapsed_hbm, elapsed_host = (elapsed_other, elapsed) if args.inputs == "host" else (elapsed, elapsed_other)

    tracker_alone_ms = None
    if args.emulate_world > 1 and world == 1 and model._bench_last_rec[0] is not None:
        # the SAME tracker work with the GPU otherwise idle: what the chain of small dependent kernels costs without the
        # detector's workgroups in its way
        from gomatching_amd.dist import unpack_records
        T_ = cfg.MODEL.TRANSFORMER
        rec, isz = model._bench_last_rec[0]
        times = []
        for _ in range(4):
            model.begin_batch([], FRAMES_PER_GPU * args.emulate_world)
            dets = unpack_records(torch.cat([rec] * args.emulate_world), isz, model.roi_heads.feature_dim, T_.NUM_POINTS)
            torch.cuda.synchronize()
            t0 = time.time()
            with torch.cuda.stream(model._tracker_stream()):    # the tracker's own lane when CUs are reserved for it
                model.track_frames(dets, 0, 0, [], new_time_cost())
            torch.cuda.synchronize()
            times.append((time.time() - t0) * 1e3)
        tracker_alone_ms = sorted(times)[len(times) // 2]

    # Roofline leg.  The timed steps replay the detector as a hipGraph, which hides individual launches from HIP
    # events; the dominant kernel is therefore bracketed with events in PROFILE_STEPS eager steps of the same
    # workload right after the timed region (same process, same buffers, same stream).  profiles/ holds the rocprofv3
    # per-kernel average over the timed (graph) steps of this command, which agrees.
    graphed = bool(model.use_graphs and model._graphs)
    model.use_graphs = False
    prof = []
    ops.set_gemm_profile(prof)                                 # HIP events around the dominant kernel's launches
    PROFILE_STEPS = 2
    tc_box[0] = new_time_cost()
    for _ in range(PROFILE_STEPS):
        pipe.push(hbm_inputs, tc_box[0])
    pipe.flush()
    barrier()
    ops.set_gemm_profile(None)
    model.use_graphs = graphed

    total_frames = FRAMES_PER_GPU * world * args.steps
    fps = total_frames / elapsed
    # north_star's decoder figure: every launch inside the six composite decoder layers that performs Q-side nn.Linear products
    # (M = frames x queries x points rows: ref_point_head, the self-attention blocks, cross-attention offsets | logits and
    # out_proj, FFN, the ctrl-point MLP) against those products' algorithmic FLOPs (7.375 GFLOP per layer and frame)
    dec_prof = [p_ for p_ in prof if len(p_) > 5 and p_[5] == "decoder_layer"]
    dec_prof = [p_ for p_ in dec_prof if not p_[4].startswith("msda:")]             # (sampling is not a Q-side product)
    ffn_prof = [p_ for p_ in prof if len(p_) > 4 and p_[4].startswith("ffn") and not p_[4].startswith("ffn-mlp2:")]   # the fused FFN block
    k256_prof = [p_ for p_ in prof if len(p_) > 4 and p_[4].startswith("k256:")]    # the decoder's row-resident K = 256 kernel
    pl_prof = [p_ for p_ in prof if len(p_) > 4 and p_[4].startswith("projln:")]    # out_proj + residual + LayerNorm launches
    all_prof = prof
    # the dominant kernel = the 128x128 tile kernel: every launch of it in the step, through the GEMM API and as the backbone's
    # pointwise convolutions ("pw:", the same instantiation: csrc/gemm_f16x3.hip dispatch<0, 0>)
    bn_prof = [p_ for p_ in prof if len(p_) > 4 and p_[4].startswith("bneck:")]     # fused bottleneck tail + next head launches
    msda_prof = [p_ for p_ in prof if len(p_) > 4 and p_[4].startswith("msda:")]    # fused multi-scale deformable attention
    c3_prof = [p_ for p_ in prof if len(p_) > 4 and p_[4].startswith("conv3:")]     # patch-resident 3x3 convolutions
    pwk_prof = [p_ for p_ in prof if len(p_) > 4 and p_[4].startswith("pwk256:")]  # 256-channel pointwise convolutions on the K = 256 kernel
    prof = [p_ for p_ in prof if not (len(p_) > 4 and p_[4].startswith(("ffn", "k256:", "projln:", "projdot:", "decattn:", "dectail:", "bneck:", "msda:", "conv3:", "pwk256:")))]
    dur_ms = sum(p[0].elapsed_time(p[1]) for p in prof)
    flops = sum(p[2] for p in prof)
    alg_bytes = sum(p[3] for p in prof)
    achieved = flops / (dur_ms * 1e-3) / 1e12 if dur_ms > 0 else 0.0
    pw_prof = [p_ for p_ in prof if len(p_) > 4 and p_[4].startswith("pw:")]
    api_prof = [p_ for p_ in prof if p_ not in pw_prof]

    def population(ps, key):
        """One population of the dominant kernel's launches against both roofs (algorithmic FLOP and bytes, HIP-event time)."""
        if not ps:
            return None
        d_ = sum(p_[0].elapsed_time(p_[1]) for p_ in ps) * 1e-3
        f_, b_ = sum(p_[2] for p_ in ps), sum(p_[3] for p_ in ps)
        return {"launches_per_step": len(ps) // PROFILE_STEPS, "avg_launch_us": d_ * 1e6 / len(ps),
                "algorithmic_bytes_per_launch_avg": b_ / len(ps), "flop_per_byte": f_ / b_,
                "hbm": {"achieved": b_ / d_ / 1e12, "peak": 8.0, "unit": "TB/s", "frac": b_ / d_ / 1e12 / 8.0},
                "mfma": {"achieved": f_ / d_ / 1e12, "peak": PEAKS[args.gemm][1], "unit": "TFLOP/s", "frac": f_ / d_ / 1e12 / PEAKS[args.gemm][1]},
                "traffic": pmc_traffic(PEAKS[args.gemm][0] + key),
                "share_of_step_time": (d_ * 1e3 / PROFILE_STEPS) / (elapsed / args.steps * 1e3)}
    if rank == 0 and os.environ.get("GOM_BENCH_WRITE_GRIDS"):   # for tools/pmc_traffic.py: work-items of the GEMM-API launches
        grids = set()
        for p_ in api_prof:
            if len(p_) > 4:
                M_, N_, _ = [int(v) for v in p_[4].split(":")[-1].split("x")]
                grids.add(((M_ + 127) // 128) * ((N_ + 127) // 128) * 256)
        with open(os.environ["GOM_BENCH_WRITE_GRIDS"], "w") as f:
            json.dump(sorted(grids), f)
    by_shape = {}
    for p_ in all_prof:
        if len(p_) > 4:
            d = by_shape.setdefault(p_[4], [0, 0.0, 0.0])
            d[0] += 1
            d[1] += p_[0].elapsed_time(p_[1])
            d[2] += p_[2]
    if rank == 0 and os.environ.get("GOM_BENCH_ALL_SHAPES"):    # diagnostic: every instrumented launch shape of a step, by time
        with open(os.environ["GOM_BENCH_ALL_SHAPES"], "w") as f:
            for k_, v_ in sorted(by_shape.items(), key=lambda kv: -kv[1][1]):
                f.write("%-44s x%-3d %9.1f us each %9.1f us per step %7.1f TFLOP/s\n" % (
                    k_, v_[0] // PROFILE_STEPS, v_[1] * 1e3 / v_[0], v_[1] * 1e3 / PROFILE_STEPS, v_[2] / (v_[1] * 1e-3) / 1e12))
    line = {
        "metric": "frames/sec (whole node), 1280x720 clip, 100 queries/frame",
        "value": fps, "unit": "frames/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPES[args.gemm], "data": "synthetic",
        "value_hbm_resident": total_frames / elapsed_hbm, "value_pcie_inclusive": total_frames / elapsed_host,
        "config": {"workload": ("configs[1]: 1280x720 ICDAR15-video clip -> %dx%d net input, %d frames/GPU, "
                                "100 queries, GoMatching_ICDAR15 (R-50 + DeepSolo + LSTMatcher, rescoring), "
                                "random-init synthetic weights" % (net_hw[0], net_hw[1], FRAMES_PER_GPU))
                   if args.backbone == "r50" else
                   ("NOT the BASELINE workload: %s backbone side measurement, %dx%d net input, %d frames/GPU, 100 "
                    "queries, DeepSolo + LSTMatcher, random-init synthetic weights"
                    % ({"swin": "Swin-T", "vitae": "ViTAEv2-S"}[args.backbone], net_hw[0], net_hw[1], FRAMES_PER_GPU)),
                   "inputs": ("resized fp32 CHW frames in pinned HOST memory when the timed window starts; the H2D copy "
                              "(%.0f MB per step) is inside the window, on an upload stream under the previous step's detector "
                              "(text_track_visualizer.py:325-334 + gom_lstmatcher.py:164-170); value_hbm_resident = same steps, "
                              "frames already in HBM" % (FRAMES_PER_GPU * 3 * net_hw[0] * net_hw[1] * 4 / 1e6))
                   if args.inputs == "host" else
                   ("resized fp32 CHW frames resident in HBM when the timed window starts (the contract's `value`); "
                    "value_pcie_inclusive = the same steps with the frames in pinned HOST memory and the H2D copy (%.0f MB per step) "
                    "inside the window, on an upload stream under the previous step's detector -- what rounds 1-5 printed as `value`"
                    % (FRAMES_PER_GPU * 3 * net_hw[0] * net_hw[1] * 4 / 1e6)),
                   "frames_per_step": FRAMES_PER_GPU * world, "emulated_world": args.emulate_world,
                   "tracker_alone_ms_per_step": tracker_alone_ms,
                   "short_term_scores": ("replicated on every rank" if args.replicate_short_term else
                                         "sharded: a rank scores the frame pairs it detected, second all-gather of the [F, 2 + nq^2] blocks")
                   if world * args.emulate_world > 1 else "single rank",
                   "long_term_match": "chain of 13 launches",
                   "tracker_cus": args.tracker_cus if args.tracker_cus >= 0 else 0,
                   "pipelining": "upload(step i+1) and detector(step i+1) overlap tracker(step i)",
                   "detector_hipgraph": graphed,
                   "parallelism": "frame-sharded dp%d + 1 all-gather/step"
                   % world if world > 1 else "single GPU",
                   "detect_frac": args.detect_frac,
                   "detections_per_frame": [len(r["instances"]) for r in res[:FRAMES_PER_GPU]],
                   "tracks": int(id_count),
                   # ranks whose records arrived through the step's one all-gather (rows of the gathered buffer / frames per
                   # rank), and the track ids of the whole clip after short-track removal (every rank holds the same)
                   "rccl_ranks_seen": (getattr(model, "_last_gathered_frames", None) or FRAMES_PER_GPU) // FRAMES_PER_GPU
                   if world > 1 else 1,
                   "collective_backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None,
                   "track_ids_per_frame": [r["instances"].track_ids.cpu().tolist() if len(r["instances"]) else []
                                           for r in res]},
        # The dominant kernel = the 128x128 tile kernel, EVERY launch of it in the step.  Its launches' mean intensity (algorithmic
        # FLOP per algorithmic byte, fp32 in / fp32 out) is below the chip's balance point, so the roof that bounds it is HBM:
        # `bound` / `achieved` / `peak` / `frac` are that view, `mfma_view` the other (the round-1 figure, 0.27, was this one on
        # the GEMM-API launches only: `api_launches.mfma`).  The two populations are also given apart, each with its own PMC traffic.
        "roofline": dict(
            ({"bound": "hbm", "achieved": alg_bytes / (dur_ms * 1e-3) / 1e12 if dur_ms > 0 else 0.0, "peak": 8.0, "unit": "TB/s",
              "frac": alg_bytes / (dur_ms * 1e-3) / 1e12 / 8.0 if dur_ms > 0 else 0.0}
             if flops < alg_bytes * PEAKS[args.gemm][1] / 8.0 else
             {"bound": "mfma", "achieved": achieved, "peak": PEAKS[args.gemm][1], "unit": "TFLOP/s", "frac": achieved / PEAKS[args.gemm][1]}),
            **{"kernel": PEAKS[args.gemm][0],
                     "mfma_view": {"bound": "mfma", "achieved": achieved, "peak": PEAKS[args.gemm][1], "unit": "TFLOP/s",
                                   "frac": achieved / PEAKS[args.gemm][1]},
                     "api_launches": population(api_prof, " [gemm api]"),
                     "pointwise_conv_launches": population(pw_prof, " [pointwise conv]"),
                     "traffic": pmc_traffic(PEAKS[args.gemm][0]), "mfma_passes_per_product": PEAKS[args.gemm][2],
                     "peak_note": {"fp32": "dense fp32-input MFMA peak",
                                   "bf16x6": "algorithmic fp32-equivalent FLOP/s; dense bf16 MFMA peak 2500 / 6 passes",
                                   "f16x3": "algorithmic fp32-equivalent FLOP/s; dense fp16 MFMA peak 2500 / 3 passes"}[args.gemm],
                     "launches_per_step": len(prof) // PROFILE_STEPS,
                     "avg_launch_us": dur_ms * 1e3 / max(len(prof), 1),
                     "flops_per_launch_avg": flops / max(len(prof), 1),
                     "algorithmic_bytes_per_launch_avg": alg_bytes / max(len(prof), 1),
                     "algorithmic_hbm_tb_per_s": alg_bytes / (dur_ms * 1e-3) / 1e12 if dur_ms > 0 else 0.0,
                     # the same launches against the OTHER roof: their mean intensity (flops / algorithmic bytes) is below the
                     # chip's balance point, so the roofline that binds them is HBM -- both views are printed, `frac` above
                     # stays the MFMA one (comparable with round 1)
                     "hbm_view": {"bound": "hbm", "achieved": alg_bytes / (dur_ms * 1e-3) / 1e12 if dur_ms > 0 else 0.0,
                                  "peak": 8.0, "unit": "TB/s",
                                  "frac": alg_bytes / (dur_ms * 1e-3) / 1e12 / 8.0 if dur_ms > 0 else 0.0,
                                  "flop_per_byte": flops / alg_bytes if alg_bytes > 0 else 0.0,
                                  "balance_flop_per_byte": PEAKS[args.gemm][1] * 1e12 / 8.0e12},
                     "hbm_note": "fp32 in, fp32 out: the K = 256 shapes of this kernel carry 44-105 FLOP per byte and the backbone's pointwise "
                                 "convolutions 32-200, around or below the chip's balance of ~104: the HBM roofline (8 TB/s) caps most "
                                 "of them at 0.3-0.5 of the MFMA peak whatever the kernel does (DESIGN.md §3)",
                     "share_of_step_time": (dur_ms / PROFILE_STEPS) / (elapsed / args.steps * 1e3),
                     "by_shape_MxNxK": {k: {"launches_per_step": v[0] // PROFILE_STEPS, "avg_us": v[1] * 1e3 / v[0],
                                            "tflops": v[2] / (v[1] * 1e-3) / 1e12,
                                            "frac": v[2] / (v[1] * 1e-3) / 1e12 / PEAKS[args.gemm][1]}
                                        for k, v in sorted(by_shape.items(), key=lambda kv: -kv[1][1])[:12]},
                     "measured_in": "%d eager steps after the timed region (timed steps are hipGraph replays)" % PROFILE_STEPS
                     if graphed else "%d eager steps after the timed region" % PROFILE_STEPS}),
        "stage_ms_per_step": {k: v / args.steps * 1e3 for k, v in tc.items() if isinstance(v, float) and v > 0},
    }
    if ffn_prof:
        fd = sum(p_[0].elapsed_time(p_[1]) for p_ in ffn_prof)
        ff = sum(p_[2] for p_ in ffn_prof)
        line["roofline_fused_ffn"] = {
            "bound": "mfma", "kernel": "ffn_fused_kernel", "achieved": ff / (fd * 1e-3) / 1e12, "peak": PEAKS["f16x3"][1],
            "unit": "TFLOP/s", "frac": ff / (fd * 1e-3) / 1e12 / PEAKS["f16x3"][1], "traffic": ffn_traffic(len(ffn_prof)),
            "launches_per_step": len(ffn_prof) // PROFILE_STEPS, "avg_launch_us": fd * 1e3 / len(ffn_prof),
            "share_of_step_time": (fd / PROFILE_STEPS) / (elapsed / args.steps * 1e3),
            "note": "linear1 + ReLU + linear2 + residual + LayerNorm of every ENCODER layer in one call: 2 KB of HBM traffic per token "
                    "instead of 13 (csrc/ffn_fused.hip).  avg_launch_us = HIP-event time of a CALL = two launches (2 304 tiles of 128 "
                    "rows, then the last round as 39 half-height tiles: ffn_fused_kernel<false, 1>).  Rounds 2-4 averaged the six "
                    "decoder calls (M = 20 000, ~85 us) into this object; from round 5 the decoder's FFN blocks run inside the tail "
                    "launch (csrc/dec_tail.hip, `roofline_decoder_qside`), so launches_per_step is 6 and avg_launch_us an encoder call's"}
        both_ms, both_fl = dur_ms + fd, flops + ff
        k256_long = [p_ for p_ in k256_prof if int(p_[4].split(":")[1].split("x")[0]) > 65536]   # encoder-sized launches
        k256_prof = [p_ for p_ in k256_prof if p_ not in k256_long]
        if k256_long:
            ld = sum(p_[0].elapsed_time(p_[1]) for p_ in k256_long)
            lf, lb = sum(p_[2] for p_ in k256_long), sum(p_[3] for p_ in k256_long)
            line["roofline_k256_long"] = {
                "bound": "hbm", "kernel": "gemm_k256_kernel<true>", "achieved": lb / (ld * 1e-3) / 1e12, "peak": 8.0, "unit": "TB/s",
                "frac": lb / (ld * 1e-3) / 1e12 / 8.0, "traffic": pmc_traffic("gemm_k256_kernel<true>"),
                "mfma_view": {"achieved": lf / (ld * 1e-3) / 1e12, "peak": PEAKS["f16x3"][1], "unit": "TFLOP/s",
                              "frac": lf / (ld * 1e-3) / 1e12 / PEAKS["f16x3"][1]},
                "launches_per_step": len(k256_long) // PROFILE_STEPS, "avg_launch_us": ld * 1e3 / len(k256_long),
                "share_of_step_time": (ld / PROFILE_STEPS) / (elapsed / args.steps * 1e3),
                "note": "the encoder's offsets | logits | value projection (N = 640, periodic position table) and the hoisted "
                        "value_proj x6 (N = 1536) at M = frames x tokens on the row-resident kernel's whole-line-store form: 86 "
                        "FLOP per HBM byte, an HBM-stream kernel (csrc/gemm_k256.hip)"}
            both_ms, both_fl = both_ms + ld, both_fl + lf
        if k256_prof:
            kd = sum(p_[0].elapsed_time(p_[1]) for p_ in k256_prof)
            kf = sum(p_[2] for p_ in k256_prof)
            line["roofline_decoder_k256"] = {
                "bound": "mfma", "kernel": "gemm_k256_kernel<false>", "achieved": kf / (kd * 1e-3) / 1e12, "peak": PEAKS["f16x3"][1],
                "unit": "TFLOP/s", "frac": kf / (kd * 1e-3) / 1e12 / PEAKS["f16x3"][1], "traffic": pmc_traffic("gemm_k256_kernel<false>"),
                "launches_per_step": len(k256_prof) // PROFILE_STEPS, "avg_launch_us": kd * 1e3 / len(k256_prof),
                "share_of_step_time": (kd / PROFILE_STEPS) / (elapsed / args.steps * 1e3),
                "note": "the decoder's Q-side nn.Linear layers (M = frames x queries x points = 20 000 rows, K = 256): rows "
                        "resident in registers, weights streamed as MFMA fragments (csrc/gemm_k256.hip); bit-identical to the "
                        "tile kernel, which is latency-bound at this M"}
            both_ms, both_fl = both_ms + kd, both_fl + kf
        if pl_prof:
            pd = sum(p_[0].elapsed_time(p_[1]) for p_ in pl_prof)
            pb = sum(p_[3] for p_ in pl_prof)
            line["roofline_proj_ln"] = {
                "bound": "hbm", "kernel": "proj_ln2_kernel<FORM> (64-row tiles, two workgroups per CU)", "achieved": pb / (pd * 1e-3) / 1e12, "peak": 8.0, "unit": "TB/s",
                "frac": pb / (pd * 1e-3) / 1e12 / 8.0, "traffic": pmc_traffic("proj_ln_kernel"),
                "launches_per_step": len(pl_prof) // PROFILE_STEPS, "avg_launch_us": pd * 1e3 / len(pl_prof),
                "share_of_step_time": (pd / PROFILE_STEPS) / (elapsed / args.steps * 1e3),
                "note": "out_proj + residual + LayerNorm of every attention block in one launch (csrc/proj_ln.hip): 3 KB of "
                        "HBM traffic per token (X, R in; Y out) instead of 5; 26 FLOP per byte, i.e. an HBM-stream kernel"}
            both_ms, both_fl = both_ms + pd, both_fl + sum(p_[2] for p_ in pl_prof)
        line["roofline"]["gemm_class_combined"] = {
            "what": "every GEMM-class launch of the step together: the tile kernel (GEMM API + pointwise convolutions), the fused FFN, "
                    "the row-resident K = 256 kernel (both forms) and out_proj + LayerNorm",
            "achieved": both_fl / (both_ms * 1e-3) / 1e12, "frac": both_fl / (both_ms * 1e-3) / 1e12 / PEAKS[args.gemm][1],
            "share_of_step_time": (both_ms / PROFILE_STEPS) / (elapsed / args.steps * 1e3)}
    if msda_prof:
        def msda_pop(ps):
            d_ = sum(p_[0].elapsed_time(p_[1]) for p_ in ps)
            b_ = sum(p_[3] for p_ in ps)
            return {"launches_per_step": len(ps) // PROFILE_STEPS, "avg_launch_us": d_ * 1e3 / len(ps),
                    "algorithmic_bytes_per_launch_avg": b_ / len(ps), "achieved": b_ / (d_ * 1e-3) / 1e12, "peak": 8.0,
                    "unit": "TB/s", "frac": b_ / (d_ * 1e-3) / 1e12 / 8.0,
                    "share_of_step_time": (d_ / PROFILE_STEPS) / (elapsed / args.steps * 1e3)}
        nqp = cfg.MODEL.TRANSFORMER.NUM_QUERIES * cfg.MODEL.TRANSFORMER.NUM_POINTS
        enc_p = [p_ for p_ in msda_prof if int(p_[4].split("x")[1]) != nqp]
        dec_p = [p_ for p_ in msda_prof if int(p_[4].split("x")[1]) == nqp]
        md = sum(p_[0].elapsed_time(p_[1]) for p_ in msda_prof)
        mb = sum(p_[3] for p_ in msda_prof)
        T_ = cfg.MODEL.TRANSFORMER
        win_on = bool(getattr(ops, "MSDA_WINDOW", False)) and args.gemm == "f16x3"
        # HBM bytes per ENCODER call from the PMC passes: the window kernel's launch + the lane kernel's tail launch (by grid)
        n_tok = int(enc_p[0][4].split("x")[1]) if enc_p else 0
        hw0 = getattr(model.detection_transformer, "_geom", None)
        n0 = 0
        for geo_ in (hw0 or {}).values():
            n0 = geo_["hw0"][0] * geo_["hw0"][1]
            if len(geo_["hw0"]) >= 4 and getattr(ops, "MSDA_WINDOW_L1", False):
                n0 += geo_["hw0"][2] * geo_["hw0"][3]
        tail_grid = ((FRAMES_PER_GPU * (n_tok - n0) + 3) // 4) * 256
        # (pmc_traffic sums the per-call bytes of every window launch of an encoder call: the level-0 and the level-1 grids)
        l1_on = bool(getattr(ops, "MSDA_WINDOW_L1", False))
        t_win, t_tail = pmc_traffic("msda_window_kernel<8,16,5,576,4,0>"), pmc_traffic("msda_fused_lanes_kernel<false> [grid %d]" % tail_grid)
        if l1_on and t_win is not None:                       # an encoder call = the level-0 launch + the level-1 launch + the tail launch
            t_l1 = pmc_traffic("msda_window_kernel<8,16,5,576,4,1>")
            t_win = None if t_l1 is None else t_win + t_l1
        enc_traffic = (t_win + t_tail) if (win_on and t_win is not None and t_tail is not None) else None
        line["roofline_msda"] = {
            "bound": "hbm", "kernel": (("msda_window_kernel<8,16,5,576,4> (level-0 queries) + <4,8,5,576,1> (level-1 queries) + "
                                        "msda_fused_lanes_kernel<false> (an encoder call = the three launches, one after the other; a "
                                        "decoder call = the lane kernel)") if getattr(ops, "MSDA_WINDOW_L1", False) else
                                       ("msda_window_kernel<8,16,5,576,4> + msda_fused_lanes_kernel<false> (an encoder call = the two "
                                        "launches, one after the other; a decoder call = the lane kernel)")) if win_on else "msda_fused_lanes_kernel<false>",
            "achieved": mb / (md * 1e-3) / 1e12, "peak": 8.0, "unit": "TB/s",
            "frac": mb / (md * 1e-3) / 1e12 / 8.0,
            "traffic": enc_traffic if win_on else pmc_traffic("msda_fused_lanes_kernel<false>"),
            "traffic_note": "HBM bytes of one ENCODER call (window kernel + the lane kernel's tail launch, rocprofv3 PMC passes)" if win_on else None,
            "launches_per_step": len(msda_prof) // PROFILE_STEPS, "avg_launch_us": md * 1e3 / len(msda_prof),
            "algorithmic_bytes_per_launch_avg": mb / len(msda_prof),
            "share_of_step_time": (md / PROFILE_STEPS) / (elapsed / args.steps * 1e3),
            # the two populations apart: the blended figure flatters the encoder launches
            "encoder_launches": msda_pop(enc_p) if enc_p else None,
            # ops.MSDA_WINDOW_POLICY: share of the window launches' octet groups that left their windows, measured once per layer on
            # its first eager call; a layer above ops.MSDA_WINDOW_MAX_FALLBACK runs on the gather kernel
            "window_fallback_share_per_encoder_layer": [model.detection_transformer.msda_window_fallback.get(i_)
                                                        for i_ in range(model.detection_transformer.n_enc)],
            "window_layers": [bool(L_.get("msda_window", True)) for L_ in model.detection_transformer.enc],
            "decoder_launches": dict(msda_pop(dec_p), note="algorithmic bytes here = the value map once + raw + output; for "
                                     "%d sparse queries per frame (512 corner lines each: %.0f MB of line reads per launch against a "
                                     "%.0f MB map) 'the map once' is an upper bound of the distinct lines touched, not a minimum -- "
                                     "read this launch's fraction as <= the printed one"
                                     % (nqp, FRAMES_PER_GPU * nqp * 512 * 128 / 1e6, FRAMES_PER_GPU * 37171 * 1024 / 1e6)) if dec_p else None,
            "note": "softmax + sampling locations + bilinear gather in one pass (csrc/msda.hip), 6 encoder + 6 decoder calls; "
                    "algorithmic bytes = value once + raw offsets | logits + output (SURVEY.md 8-d); avg_launch_us = HIP-event time "
                    "of a CALL.  Round 4: an encoder call serves its level-0 queries (75 %) from per-workgroup LDS windows of the "
                    "value map (msda_window_kernel: ~10 lines fetched per (query, head) instead of 64 gathered; bound by the vector "
                    "ALU, tools/exp/msda_window_clock.py) and runs the coarser levels' queries on the lane kernel behind it; the lane "
                    "kernel alone is bound by the texture-address path (512 corner lines per query, TA busy 0.965, "
                    "profiles/r03_msda_ta_counters.txt; no cheaper gather shape or layout: profiles/r04_msda_ta_counters.txt)"}
    if bn_prof:
        bd = sum(p_[0].elapsed_time(p_[1]) for p_ in bn_prof)
        bb, bf = sum(p_[3] for p_ in bn_prof), sum(p_[2] for p_ in bn_prof)
        line["roofline_bneck"] = {
            "bound": "hbm", "kernel": "bneck_kernel<K1,MP,OCC> (res2, res3) + bneck2_kernel<K1,256> (res4 and the res3 -> res4 transition)", "achieved": bb / (bd * 1e-3) / 1e12, "peak": 8.0, "unit": "TB/s",
            "frac": bb / (bd * 1e-3) / 1e12 / 8.0, "traffic": pmc_traffic("bneck_kernel"),
            "mfma_view": {"achieved": bf / (bd * 1e-3) / 1e12, "peak": PEAKS["f16x3"][1], "unit": "TFLOP/s",
                          "frac": bf / (bd * 1e-3) / 1e12 / PEAKS["f16x3"][1]},
            "launches_per_step": len(bn_prof) // PROFILE_STEPS, "avg_launch_us": bd * 1e3 / len(bn_prof),
            "share_of_step_time": (bd / PROFILE_STEPS) / (elapsed / args.steps * 1e3),
            "note": "conv3 + BN + residual + ReLU of a ResNet bottleneck block and conv1 + BN + ReLU of the next in one launch "
                    "(res2 / res3: csrc/bneck_fused.hip): the block's output is written once and not read back by conv1; 26-50 "
                    "FLOP per byte, an HBM-stream kernel"}
    if pwk_prof:
        wd_ = sum(p_[0].elapsed_time(p_[1]) for p_ in pwk_prof)
        wb_, wf_ = sum(p_[3] for p_ in pwk_prof), sum(p_[2] for p_ in pwk_prof)
        line["roofline_pw_k256"] = {
            "bound": "hbm", "kernel": "gemm_k256_kernel<false> (res4 conv3: 256 -> 1024 + BN + shortcut + ReLU)",
            "achieved": wb_ / (wd_ * 1e-3) / 1e12, "peak": 8.0, "unit": "TB/s", "frac": wb_ / (wd_ * 1e-3) / 1e12 / 8.0,
            "mfma_view": {"achieved": wf_ / (wd_ * 1e-3) / 1e12, "peak": PEAKS["f16x3"][1], "unit": "TFLOP/s",
                          "frac": wf_ / (wd_ * 1e-3) / 1e12 / PEAKS["f16x3"][1]},
            "launches_per_step": len(pwk_prof) // PROFILE_STEPS, "avg_launch_us": wd_ * 1e3 / len(pwk_prof),
            "share_of_step_time": (wd_ / PROFILE_STEPS) / (elapsed / args.steps * 1e3)}
    if c3_prof:
        cd_ = sum(p_[0].elapsed_time(p_[1]) for p_ in c3_prof)
        cb_, cf_ = sum(p_[3] for p_ in c3_prof), sum(p_[2] for p_ in c3_prof)
        line["roofline_conv3x3"] = {
            "bound": "mfma", "kernel": "conv3x3_patch_kernel<BN>", "achieved": cf_ / (cd_ * 1e-3) / 1e12, "peak": PEAKS["f16x3"][1],
            "unit": "TFLOP/s", "frac": cf_ / (cd_ * 1e-3) / 1e12 / PEAKS["f16x3"][1], "traffic": pmc_traffic("conv3x3_patch_kernel"),
            "hbm_view": {"achieved": cb_ / (cd_ * 1e-3) / 1e12, "peak": 8.0, "unit": "TB/s", "frac": cb_ / (cd_ * 1e-3) / 1e12 / 8.0},
            "launches_per_step": len(c3_prof) // PROFILE_STEPS, "avg_launch_us": cd_ * 1e3 / len(c3_prof),
            "share_of_step_time": (cd_ / PROFILE_STEPS) / (elapsed / args.steps * 1e3),
            "note": "conv2 (3x3 / 1) of the ResNet bottlenecks with the input patch resident in LDS and the weights streamed by "
                    "LDS-DMA (csrc/conv3x3_patch.hip): 76-81 % matrix-pipe busy in its loop at the 1.4-1.6 GHz the chip holds under "
                    "it (tools/exp/conv3_clock.py); the implicit-GEMM kernel it replaces ran these launches at ~0.30"}
    if dec_prof:
        T_ = cfg.MODEL.TRANSFORMER
        rows = FRAMES_PER_GPU * T_.NUM_QUERIES * T_.NUM_POINTS
        d_, F_, L_ = T_.HIDDEN_DIM, T_.DIM_FEEDFORWARD, T_.DEC_LAYERS
        # per layer: ref_point_head 2 x (d x d), in_proj 3d + out_proj d (twice), offsets | logits 384 + out_proj d, FFN 2 x d x F,
        # ctrl-point MLP 2 x (d x d) + d x 2
        qside = 2.0 * rows * d_ * (2 * d_ + 2 * 4 * d_ + 384 + d_ + 2 * F_ + 2 * d_ + 2) * L_
        dd = sum(p_[0].elapsed_time(p_[1]) for p_ in dec_prof) / PROFILE_STEPS * 1e-3
        def pipe_busy(kind, us_per_launch):
            """Static MFMA work of the decoder's launches per SIMD (see csrc/dec_attn.hip, dec_attn2.hip, dec_tail.hip) against their
            duration.  (dec_attn2: two 16-token waves per SIMD on the 16-cycle shape issue what one 32-token wave issues on the 32-cycle
            shape: the same cycles per SIMD in both forms.)"""
            # the tail launch by the form in effect (ops.DEC_TAIL2, ops.DEC_TAIL_PROJ): form 2 = 80-row workgroups, a wave issues 480
            # MFMAs per 128-unit chunk and per plain 256 -> 256 layer; form 1 = 128-row workgroups, 192 per 32-unit stage
            tail2 = bool(getattr(ops, "DEC_TAIL2", False)) and F_ % 128 == 0
            with_proj = 1 if getattr(ops, "DEC_TAIL_PROJ", False) else 0
            tail_cyc = (with_proj + F_ // 128 + 2 + 2 * (L_ - 1) / L_) * 480 * 16 if tail2 else \
                (4 * with_proj + F_ // 32 + 8 + 8 * (L_ - 1) / L_) * 192 * 16         # (the last layer's launch has no ref_point_head block)
            wgs = {"decattn intra": (rows // T_.NUM_POINTS + 3) // 4, "decattn inter": FRAMES_PER_GPU * T_.NUM_POINTS,
                   "decattn inter+raw": FRAMES_PER_GPU * T_.NUM_POINTS, "dectail": (rows + 79) // 80 if tail2 else (rows + 127) // 128}
            cyc = {"decattn intra": (32 * 48 + 8 * 12) * 32, "decattn inter": (32 * 48 + 8 * 48) * 32,
                   "decattn inter+raw": (44 * 48 + 8 * 48) * 32, "dectail": tail_cyc}
            if kind not in cyc or us_per_launch <= 0:
                return {}
            return {"mfma_cycles_per_wave": cyc[kind], "matrix_pipe_busy": cyc[kind] / (us_per_launch * 2400.0),
                    "cus_with_work": min(1.0, wgs[kind] / 256.0)}
        kinds = {}
        for p_ in dec_prof:
            k_ = p_[4].split(":")[0] if ":" in p_[4] else ("ffn" if p_[4].startswith("ffn") else "tile / fp32 GEMM")
            if p_[4].startswith("ffn-mlp2:"):
                k_ = "two-layer perceptron (ffn_fused, PLAIN)"
            if p_[4].startswith("decattn:"):
                k_ = "decattn " + p_[4].split(":")[1]
            e_ = kinds.setdefault(k_, [0, 0.0])
            e_[0] += 1
            e_[1] += p_[0].elapsed_time(p_[1])
        line["roofline_decoder_qside"] = {
            "bound": "mfma", "achieved": qside / dd / 1e12, "peak": PEAKS["f16x3"][1], "unit": "TFLOP/s",
            "frac": qside / dd / 1e12 / PEAKS["f16x3"][1],
            "algorithmic_gflop_per_frame": qside / FRAMES_PER_GPU / 1e9, "launches_per_step": len(dec_prof) // PROFILE_STEPS,
            "us_per_step": dd * 1e6, "share_of_step_time": dd * 1e3 / (elapsed / args.steps * 1e3),
            "by_kernel_us_per_step": {k_: dict({"launches": v_[0] // PROFILE_STEPS, "us": v_[1] * 1e3 / PROFILE_STEPS},
                                           **pipe_busy(k_, v_[1] * 1e3 / max(v_[0], 1)))
                                      for k_, v_ in sorted(kinds.items())},
            "pipe_busy_note": "mfma_cycles_per_wave = the launch's MFMA instructions per SIMD (static count, csrc: one wave per SIMD, or the "
                              "two 16-token waves of dec_attn2) x their pipe cycles (32 for 32x32x16, 16 for 16x16x32); matrix_pipe_busy = that over the launch's duration at the nominal 2.4 GHz "
                              "(a LOWER bound of the busy fraction of a SIMD that holds a wave: the chip clocks at or below 2.4 GHz); "
                              "cus_with_work = one-per-CU workgroups of the launch / 256.  The chip fraction of a launch is the product",
            "note": "the six composite decoder layers' Q-side nn.Linear products (44.25 GFLOP per frame at 100 queries) over the "
                    "summed HIP-event time of EVERY launch that performs one of them -- the fused self-attention blocks "
                    "(csrc/dec_attn.hip) are counted whole, attention cores and LayerNorms included; MSDA sampling, the sine "
                    "embedding and ref_sigmoid launches are not Q-side products and are left out.  north_star asks 0.60"}
    # `roofline` (the contract's object) = the kernel with the largest share of the step among the instrumented ones, measured
    # in this run; rounds 1-2 it was the 128x128 tile kernel, whose object now lives on as `roofline_tile_gemm`
    line["roofline_tile_gemm"] = line.pop("roofline")
    shares = {k_: line[k_].get("share_of_step_time", 0.0)
              for k_ in ("roofline_tile_gemm", "roofline_fused_ffn", "roofline_msda", "roofline_k256_long", "roofline_proj_ln",
                         "roofline_bneck", "roofline_conv3x3") if k_ in line}
    top = max(shares, key=shares.get)

    def contract_view(k_):
        return {"bound": line[k_]["bound"], "achieved": line[k_]["achieved"], "peak": line[k_]["peak"],
                "unit": line[k_]["unit"], "frac": line[k_]["frac"], "traffic": line[k_].get("traffic"),
                "kernel": line[k_].get("kernel"), "same_as": k_, "share_of_step_time": shares[k_],
                "launches_per_step": line[k_].get("launches_per_step"), "avg_launch_us": line[k_].get("avg_launch_us"),
                "shares_of_step_time": shares, "measured_in": line["roofline_tile_gemm"].get("measured_in")}
    # The contract's `roofline` object is PINNED to one kernel from round 5 on -- the fused FFN (it and MSDA trade the largest share
    # between boxes, 0.2 % of a step apart: rounds 1-4 named tile GEMM, tile GEMM, MSDA, FFN) -- so that `roofline.frac` compares like
    # for like across rounds; `roofline_top` is the largest-share view of THIS run, whichever kernel that is.
    pinned = "roofline_fused_ffn" if "roofline_fused_ffn" in shares else top
    line["roofline"] = dict(contract_view(pinned), pinned=True)
    # the other two figures a reader of the contract object asks for first (VERDICT r5 "small" 8): north_star's decoder figure
    # and the MSDA call that holds the largest share; the full objects are `roofline_decoder_qside` / `roofline_msda`
    dq, ms_ = line.get("roofline_decoder_qside") or {}, line.get("roofline_msda") or {}
    line["roofline"]["also"] = {
        "decoder_qside_frac_of_mfma_peak": dq.get("frac"), "decoder_qside_us_per_step": dq.get("us_per_step"),
        "decoder_qside_launch_us": {k_: (v_["us"] / v_["launches"]) for k_, v_ in (dq.get("by_kernel_us_per_step") or {}).items()
                                    if v_.get("launches")},
        "msda_encoder_call": ms_.get("encoder_launches"), "msda_share_of_step_time": ms_.get("share_of_step_time"),
        "msda_window_fallback_share_per_encoder_layer": ms_.get("window_fallback_share_per_encoder_layer"),
        "in_kernel_clock_note": "fractions are against peaks at 2.4 GHz; under its MFMA-dense kernels the chip holds 1.4-2.1 GHz "
                                "(profiles/r05_clock_stamps.log, profiles/r06_lds_rate.log: 1.76-1.88 GHz in bare MFMA + LDS loops)"}
    line["roofline_top"] = contract_view(top)
    solo = rank == 0 and world == 1 and args.backbone == "r50" and args.emulate_world == 1
    line["fallback_steps"] = int(model.fallback_steps)          # steps of this run re-done on the bf16x6 twin (range flag tripped)
    assert model.fallback_steps == 0 or args.gemm != "f16x3", "an f16x3 step fell back to bf16x6 inside the measurement"
    legs = [] if (args.no_config_legs or not solo or args.config == "ic15") else (["dstext", "bovtext"] if args.config == "all" else [args.config])
    if legs:
        del pipe
        model._graphs.clear()                                    # the captured graph pins the headline model's activation pool
        torch.cuda.empty_cache()
    for leg in legs:
        line["value_" + leg] = config_leg(leg, device, args.gemm, args.detect_frac)
    if solo and not args.no_alt_backends:
        # the other two contraction back-ends on the same window, a few steps each (secondary figures, same process)
        alt = {}
        for mode in [m for m in ("f16x3", "bf16x6", "fp32") if m != args.gemm]:
            m2, _, pipe2, tcb2 = make(mode)
            setup(pipe2, tcb2, primary)
            e2, _, _ = timed(pipe2, tcb2, primary, 4, 1)
            alt[mode] = {"value": FRAMES_PER_GPU * 4 / e2, "ms_per_step": e2 / 4 * 1e3, "steps": 4}
            del m2, pipe2, tcb2
            torch.cuda.empty_cache()
        ops.GEMM_MODE = args.gemm
        line["alt_backends"] = alt
    if solo and not args.no_cpu_baseline:
        cpu_cfg = setup_cfg(builtin="icdar15")
        cpu_cfg.MODEL.DEVICE = "cpu"
        line["cpu_baseline"] = cpu_baseline(cpu_cfg, sd, shifts["s"], shifts["r"], [x["image"] for x in host_inputs],
                                            hw, res, id_count)
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
