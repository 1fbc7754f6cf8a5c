"""How long the (replicated) tracker takes as the clip grows: the N-GPU bench tracks 8N frames per step on every rank,
so its time against one step of detection (8 frames) bounds multi-GPU scaling.  Detects 64 frames once, then times
track_frames + short-track removal + postprocess over the first 8/16/32/64 of them (through the same record
pack/unpack the all-gather path uses)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gomatching_amd.config import setup_cfg  # noqa: E402
from gomatching_amd.dist import pack_records, unpack_records  # noqa: E402
from gomatching_amd.predictor import GoMBatchPredictor, new_time_cost  # noqa: E402
from gomatching_amd.synth import make_clip  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = setup_cfg(builtin="icdar15")
    cfg.MODEL.DEVICE = "cuda"
    model, sd = bench.build_model(cfg, dev)
    predictor = GoMBatchPredictor(cfg, model)
    clip = make_clip(64, 720, 1280, clip_id=0, num_rects=12)
    inputs, hw = predictor.prepare([f[:, :, ::-1] for f in clip])
    inputs = [dict(x, image=x["image"].to(dev)) for x in inputs]
    bench.calibrate(model, inputs[:1])
    T = cfg.MODEL.TRANSFORMER
    tc = new_time_cost()
    recs = []
    t0 = time.time()
    for s in range(0, 64, 8):
        model.begin_batch([], 8)
        dets = model.inference(inputs[s:s + 8], tc)
        recs.append(pack_records(dets, T.NUM_QUERIES, model.roi_heads.feature_dim, T.NUM_POINTS, dev))
    torch.cuda.synchronize()
    print("detection: %.1f ms per 8-frame step (unpipelined)" % ((time.time() - t0) / 8 * 1e3))
    allrec = torch.cat(recs)
    size = dets[0].image_size
    for n in (8, 16, 32, 64, 64):
        tc = new_time_cost()
        torch.cuda.synchronize()
        t0 = time.time()
        model.begin_batch([], n)
        all_dets = unpack_records(allrec[:n], size, model.roi_heads.feature_dim, T.NUM_POINTS)
        t1 = time.time()
        if "--profile-track" in sys.argv and n == 64:
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            insts, id_count = model.track_frames(all_dets, 0, 0, [], tc)
            pr.disable()
            pstats.Stats(pr).sort_stats("tottime").print_stats(22)
        else:
            insts, id_count = model.track_frames(all_dets, 0, 0, [], tc)
        t2 = time.time()
        insts = model._remove_short_track(insts)
        res = model.batch_postprocess(insts, [hw] * len(insts))
        torch.cuda.synchronize()
        t3 = time.time()
        nd = sum(len(d) for d in all_dets)
        print("frames %2d  dets %4d  tracks %4d | unpack %.1f ms  track %.1f ms (short %.1f long %.1f)  post %.1f ms | "
              "total %.1f ms = %.2f ms/frame" % (n, nd, id_count, (t1 - t0) * 1e3, (t2 - t1) * 1e3,
                                                 tc["short_match"] * 1e3, tc["long_match"] * 1e3, (t3 - t2) * 1e3,
                                                 (t3 - t0) * 1e3, (t3 - t0) * 1e3 / n))


if __name__ == "__main__":
    if "--profile" in sys.argv:
        import cProfile
        import pstats
        cProfile.run("main()", "/tmp/tracker.prof")
        pstats.Stats("/tmp/tracker.prof").sort_stats("tottime").print_stats(45)
    else:
        main()
