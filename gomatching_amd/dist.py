"""Frame-sharded inference across the GPUs of one node (the reference has no
inference-time parallelism, eval.py:297-345).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" for CPU tests).
A clip of F*world frames is split into contiguous blocks of F frames per rank.  Detection + embedding
(A1-A13) is embarrassingly parallel; the tracker (A14-A17) is a serial recurrence over frames, so after ONE
`all_gather_into_tensor` of fixed-shape per-frame association records every rank holds identical inputs
and runs the tracker replicated (deterministic => no broadcast of ids needed).

Record layout, fp32, per frame: [nq + 1, D] with D = F_reid + 4 + 1 + 2P + 4P + P
    row 0       : [count, image height, image width, 0, ...]   (frames of different ranks may differ in size: BASELINE config 5)
    rows 1..nq  : reid[F_reid] | box[4] | score | ctrl[2P] | bd[4P] | recs[P]   (first `count` rows valid)
= 0.48 MB/frame at nq=100 (1.45 MB at nq=300): the exchange is latency-bound on xGMI, hence one fused
buffer per step rather than per-field collectives.
"""
import numpy as np
import torch

from .structures import Boxes, Instances


def record_dim(feature_dim, num_points):
    return feature_dim + 4 + 1 + 2 * num_points + 4 * num_points + num_points


def pack_records(dets, nq, feature_dim, num_points, device):
    """dets: list of per-frame Instances from GoMatching.inference -> [F, nq+1, D] fp32 on `device`.
    Frames that come straight out of `GoMatching.detect_finish` (one step: nq-padded device arrays + consecutive pool rows)
    are packed by ONE kernel (csrc/records.hip); anything else takes the generic per-field path."""
    P = num_points
    D = record_dim(feature_dim, P)
    g0 = getattr(dets[0], "_gom", None) if dets else None
    det = g0.get("det") if g0 else None
    if det is not None and det["ctrl"].is_cuda and all(
            getattr(r, "_gom", None) is not None and r._gom.get("det") is det and r._gom.get("b") == b
            and r.image_size == dets[0].image_size for b, r in enumerate(dets)) and len(dets) == det["count"].numel():
        from . import ops
        return ops.pack_records(g0["pool"], g0["row0"], det, len(dets), nq, feature_dim, P, dets[0].image_size)
    buf = torch.zeros((len(dets), nq + 1, D), dtype=torch.float32, device=device)
    for f, r in enumerate(dets):
        n = len(r)
        buf[f, 0, 0] = float(n)
        buf[f, 0, 1], buf[f, 0, 2] = float(r.image_size[0]), float(r.image_size[1])
        if n == 0:
            continue
        o = 0
        row = buf[f, 1:n + 1]
        for field, width in ((r.reid_features, feature_dim), (r.pred_boxes.tensor, 4), (r.scores.view(n, 1), 1),
                             (r.ctrl_points.reshape(n, 2 * P), 2 * P), (r.bd.reshape(n, 4 * P), 4 * P),
                             (r.recs.reshape(n, P).to(torch.float32), P)):
            row[:, o:o + width] = field
            o += width
    return buf


def unpack_records(buf, image_size, feature_dim, num_points):
    """[F, nq+1, D] -> list of Instances (device tensors are views into `buf` / into per-call bulk conversions, so
    the number of kernels does not grow with F) with host mirrors attached.  Every frame takes its image size from its own
    record header (`image_size` is only the fall-back for records written without one)."""
    P = num_points
    hdr = buf[:, 0, :3].cpu().numpy()
    counts = hdr[:, 0].astype(np.int64)
    small = buf[:, 1:, feature_dim:feature_dim + 5].cpu().numpy()          # boxes + score, one D2H
    o_rec = feature_dim + 5 + 6 * P
    recs_all = buf[:, 1:, o_rec:o_rec + P].to(torch.int64)                  # one conversion for every frame
    nmax = int(counts.max()) if len(counts) else 0
    classes = torch.zeros((max(nmax, 1),), dtype=torch.int64, device=buf.device)
    out = []
    for f in range(buf.shape[0]):
        n = int(counts[f])
        row = buf[f, 1:n + 1]
        o = feature_dim
        hw = (int(hdr[f, 1]), int(hdr[f, 2])) if hdr[f, 1] > 0 and hdr[f, 2] > 0 else image_size
        r = Instances(hw)
        r.reid_features = row[:, :feature_dim]
        r.pred_boxes = Boxes(row[:, o:o + 4])
        r.scores = row[:, o + 4]
        r.pred_classes = classes[:n]
        o += 5
        r.ctrl_points = row[:, o:o + 2 * P]
        o += 2 * P
        r.bd = row[:, o:o + 4 * P].reshape(n, P, 4)
        r.recs = recs_all[f, :n]
        r._gom = {"boxes": small[f, :n, :4].copy(), "scores": small[f, :n, 4].copy(), "row0": None, "ids": None}
        out.append(r)
    return out


def all_gather_records(local, group=None, always_collective=False):
    """One fused collective per step: [F,nq+1,D] on every rank -> [world*F, nq+1, D] in rank order.  `always_collective`:
    issue the collective at world size 1 too (tests: the RCCL communicator and call path on the one GPU a test box has)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1 and not always_collective:
        return local
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # test configuration only (several ranks sharing one GPU): gloo moves host memory
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local.cpu().contiguous(), group=group)
        out.copy_(host)
        return out
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def sharded_batch_inference(model, local_inputs, batch_id, id_count, instances, time_cost, group=None):
    """`GoMatching.batch_inference` for a clip whose frames are block-sharded over the ranks: returns the ids
    of ALL world*F frames on every rank."""
    import torch.distributed as dist
    T = model.cfg.MODEL.TRANSFORMER
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return model.batch_inference(local_inputs, batch_id, id_count, instances, time_cost)
    model.begin_batch(instances, len(local_inputs) * world)
    dets = model.detect_steps(local_inputs, time_cost)
    return exchange_and_track(model, dets, batch_id, id_count, instances, time_cost, group)


def pack_short_term(st, t_list, nq, device="cpu"):
    """The short-term score blocks of a rank's frame pairs as one fixed-shape buffer for the second all-gather:
    [len(t_list), 2 + nq * nq] fp32, row j = (n_cur, n_prev | S of pair t_list[j], row-major, zero-padded); a pair without a
    block (an empty frame, the clip's first frame) has n_cur = n_prev = 0."""
    buf = np.zeros((len(t_list), 2 + nq * nq), np.float32)
    for j, t in enumerate(t_list):
        S = st.get(t)
        if S is None:
            continue
        n_cur, n_prev = S.shape
        assert n_cur <= nq and n_prev <= nq
        buf[j, 0], buf[j, 1] = n_cur, n_prev
        buf[j, 2:2 + n_cur * n_prev] = np.asarray(S, np.float32).reshape(-1)
    return torch.from_numpy(buf).to(device)


def unpack_short_term(buf, t_all):
    """[len(t_all), 2 + nq * nq] (rank order = frame order) -> {t: S numpy [n_cur, n_prev]} as `precompute_short_term` returns it."""
    host = buf.cpu().numpy()
    out = {}
    for j, t in enumerate(t_all):
        n_cur, n_prev = int(host[j, 0]), int(host[j, 1])
        if n_cur and n_prev:
            out[t] = host[j, 2:2 + n_cur * n_prev].reshape(n_cur, n_prev).copy()
    return out


SHARD_SHORT_TERM = True      # every rank scores only its own frame pairs; a second, small all-gather exchanges the blocks


def exchange_and_track(model, dets, batch_id, id_count, instances, time_cost, group=None, shard_short_term=None):
    """Second half of the sharded step: pack this rank's detections, ONE all-gather, replicated tracker.  The tracker's
    id-independent device work -- the short-term association scores of every consecutive frame pair
    (`precompute_short_term`, gom_lstmatcher.py:405-465 up to the assignment) -- is SHARDED: rank r scores the pairs whose
    current frame it detected (the embeddings of both frames are in the gathered buffer), and a second all-gather of the
    [F, 2 + nq^2] blocks (40 KB per frame at 100 queries) gives every rank all of them; the serial id recurrence
    (short-term assignment, long-term matches: :467-564) then runs replicated on identical inputs."""
    import torch.distributed as dist
    T = model.cfg.MODEL.TRANSFORMER
    hw = dets[0].image_size
    F_local = len(dets)
    rec = pack_records(dets, T.NUM_QUERIES, model.roi_heads.feature_dim, T.NUM_POINTS, model.device)
    allrec = all_gather_records(rec, group)
    model._last_gathered_frames = int(allrec.shape[0])         # bench.py reports the ranks seen in the gathered buffer
    all_dets = unpack_records(allrec, hw, model.roi_heads.feature_dim, T.NUM_POINTS)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    shard = SHARD_SHORT_TERM if shard_short_term is None else shard_short_term
    if world == 1 or not shard:
        return model.track_frames(all_dets, batch_id, id_count, instances, time_cost)
    rank = dist.get_rank(group)
    # fixed-shape exchange: every rank contributes the same number of frames (the block-sharding of a clip: bench.py, north_star)
    assert int(allrec.shape[0]) == world * F_local, ("ranks detected different numbers of frames", int(allrec.shape[0]), world, F_local)
    import time
    t0 = time.time()
    base = 1 if len(instances) else 0
    window = ([instances[-1]] if base else []) + list(all_dets)
    carried = list(instances[-max(model.test_len - 1, 1):]) if base else []
    model._home_features(carried + list(all_dets))             # (what track_frames does first: the pool rows of every frame)
    mine = [base + rank * F_local + j for j in range(F_local)]  # indices into `window` of the frames this rank detected
    st_local = model.precompute_short_term(window, only=set(mine))
    blocks = pack_short_term(st_local, mine, T.NUM_QUERIES, model.device)
    allblk = all_gather_records(blocks, group)
    st = unpack_short_term(allblk, [base + j for j in range(world * F_local)])
    time_cost["short_match"] = time_cost.get("short_match", 0.0) + (time.time() - t0)   # this rank's share + the second exchange
    return model.track_frames(all_dets, batch_id, id_count, instances, time_cost, st=st)
