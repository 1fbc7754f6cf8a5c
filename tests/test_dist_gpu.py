"""GPU: the frame-sharded path end to end with two ranks sharing cuda:0 (gloo carries the records in this
test; the product uses RCCL): every rank must reproduce the single-process track ids of the whole clip."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _inputs(hw, frames):
    from gomatching_amd.synth import make_clip
    clip = make_clip(frames, hw[0], hw[1], clip_id=1)
    return [{"image": torch.as_tensor(f.astype("float32").transpose(2, 0, 1))} for f in clip]


def _model():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import mini_cfg, golden, e2e_state_dict
    from gomatching_amd.modeling import GoMatching
    g = golden("e2e_lst.npz")
    cfg = mini_cfg("icdar15", device="cuda")
    return GoMatching(cfg, e2e_state_dict(cfg, g), device="cuda:0", frames_per_step=4), g


def _tc():
    return {k: 0.0 for k in ("pre_process", "backbone", "detector", "rescore", "tracker", "short_match", "long_match")}


def _worker(rank, world, port, q, shard=True):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gomatching_amd import dist as gdist
        from gomatching_amd.dist import sharded_batch_inference
        gdist.SHARD_SHORT_TERM = shard                        # True: every rank scores its own frame pairs + a second all-gather
        model, g = _model()
        hw = tuple(int(v) for v in g["hw"])
        inputs = _inputs(hw, 8)
        local = inputs[rank * 4:(rank + 1) * 4]
        insts, id_count = sharded_batch_inference(model, local, 0, 0, [], _tc())
        q.put((rank, int(id_count), [x.track_ids.cpu().tolist() for x in insts],
               [np.round(x.scores.cpu().numpy(), 5).tolist() for x in insts]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shard", [True, False])
def test_sharded_equals_single_process(shard):
    """shard: the short-term scores of a rank's own frame pairs only + the second all-gather (default) | every rank scores every
    pair (the replicated tracker of rounds 1-4).  Identical ids either way."""
    model, g = _model()
    hw = tuple(int(v) for v in g["hw"])
    insts, id_count = model.batch_inference(_inputs(hw, 8), 0, 0, [], _tc())
    ref_ids = [x.track_ids.cpu().tolist() for x in insts]
    assert ref_ids == [g["pre_ids_%d" % f].tolist() for f in range(8)]          # and both equal the reference's
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, shard)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    for rank, idc, ids, scores in res:
        assert idc == int(id_count), (rank, idc, id_count)
        assert ids == ref_ids, rank


def test_pack_kernel_equals_generic_path_and_carries_image_size():
    """A step straight out of detect_finish is packed by ONE kernel (csrc/records.hip); the result must equal the generic
    per-field path bit for bit, and every frame's record must carry its own image size (mixed-size ranks, BASELINE config 5)."""
    from gomatching_amd import dist as gdist
    model, g = _model()
    hw = tuple(int(v) for v in g["hw"])
    T = model.cfg.MODEL.TRANSFORMER
    model.begin_batch([], 4)
    dets = model.detect_steps(_inputs(hw, 4), _tc())
    assert all(d._gom.get("det") is dets[0]._gom["det"] for d in dets)
    fast = gdist.pack_records(dets, T.NUM_QUERIES, model.roi_heads.feature_dim, T.NUM_POINTS, model.device)
    for d in dets:                                            # hide the step handle: generic path
        d._gom = dict(d._gom, det=None)
    slow = gdist.pack_records(dets, T.NUM_QUERIES, model.roi_heads.feature_dim, T.NUM_POINTS, model.device)
    torch.cuda.synchronize()
    assert sum(len(d) for d in dets) > 0
    assert torch.equal(fast, slow)
    back = gdist.unpack_records(fast, (1, 1), model.roi_heads.feature_dim, T.NUM_POINTS)
    assert all(b.image_size == hw for b in back)
    assert [len(b) for b in back] == [len(d) for d in dets]


def _rccl_worker(port, q):
    """Always answers: ('ok', ...) or ('error', traceback) -- a failed RCCL init must not leave the parent waiting."""
    import traceback
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        try:
            from gomatching_amd.dist import all_gather_records, exchange_and_track, pack_records
            model, g = _model()
            hw = tuple(int(v) for v in g["hw"])
            inputs = _inputs(hw, 8)
            single, count = model.batch_inference(inputs, 0, 0, [], _tc())
            want = [x.track_ids.cpu().tolist() for x in single]
            T = model.cfg.MODEL.TRANSFORMER
            model.begin_batch([], 8)
            dets = model.detect_steps(inputs, _tc())
            rec = pack_records(dets, T.NUM_QUERIES, model.roi_heads.feature_dim, T.NUM_POINTS, model.device)
            out = all_gather_records(rec, always_collective=True)                 # ncclAllGather on the record buffer, RCCL
            torch.cuda.synchronize()
            same = bool(torch.equal(out, rec)) and out.data_ptr() != rec.data_ptr()
            model.begin_batch([], 8)
            dets = model.detect_steps(inputs, _tc())
            insts, count2 = exchange_and_track(model, dets, 0, 0, [], _tc())      # the sharded step's second half under backend nccl
            got = [x.track_ids.cpu().tolist() for x in insts]
            q.put(("ok", same, dist.get_backend(), got == want and int(count2) == int(count)))
        finally:
            dist.destroy_process_group()
    except BaseException:
        q.put(("error", traceback.format_exc()))
        raise


def test_rccl_communicator_and_all_gather_at_world_size_one():
    """What one GPU allows of the RCCL path: a `nccl` process group (= RCCL on ROCm) of world size 1, `all_gather_into_tensor`
    of the per-frame record buffer through it, and `exchange_and_track` under that backend -- communicator creation, the
    collective's launch on the record layout and the stream ordering behind the pack kernel are exercised on hardware; the
    exchange between ranks over xGMI is not (no multi-GPU box)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(port, q))
    p.start()
    import queue
    try:
        msg = q.get(timeout=300)
    except queue.Empty:
        p.join(5)
        raise AssertionError("the RCCL worker sent nothing in 300 s (exit code %r)" % (p.exitcode,))
    p.join(60)
    assert msg[0] == "ok", "the RCCL worker failed:\n" + str(msg[1])
    _, same, backend, ids_ok = msg
    assert p.exitcode == 0, p.exitcode
    assert backend == "nccl" and same and ids_ok
