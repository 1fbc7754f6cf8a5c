"""CPU, world_size 2 over gloo: the multi-GPU exchange layer (record packing, one all_gather_into_tensor per
step, unpacking) gives every rank identical, correctly ordered per-frame association inputs."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gomatching_amd.dist import pack_records, unpack_records, all_gather_records, record_dim, pack_short_term, unpack_short_term
from gomatching_amd.structures import Boxes, Instances

NQ, F, P = 12, 1024, 25


def _dets(rank, frames=3):
    g = torch.Generator().manual_seed(100 + rank)
    out = []
    for f in range(frames):
        n = [5, 0, NQ][f % 3] if rank == 0 else [1, 7, 3][f % 3]
        r = Instances((96, 128) if rank == 0 else (130, 90))         # ranks may hold frames of different sizes
        r.reid_features = torch.rand(n, F, generator=g)
        r.pred_boxes = Boxes(torch.rand(n, 4, generator=g) * 90)
        r.scores = torch.rand(n, generator=g)
        r.pred_classes = torch.zeros(n, dtype=torch.int64)
        r.ctrl_points = torch.rand(n, 2 * P, generator=g) * 100
        r.recs = torch.randint(0, 5462, (n, P), generator=g)
        r.bd = torch.rand(n, P, 4, generator=g) * 100
        out.append(r)
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        local = _dets(rank)
        rec = pack_records(local, NQ, F, P, "cpu")
        assert rec.shape == (3, NQ + 1, record_dim(F, P))
        allrec = all_gather_records(rec)
        got = unpack_records(allrec, (96, 128), F, P)
        ok = len(got) == 3 * world
        for r in range(world):
            exp = _dets(r)
            for f in range(3):
                a, b = got[r * 3 + f], exp[f]
                ok &= len(a) == len(b)
                ok &= a.image_size == b.image_size
                ok &= torch.equal(a.reid_features, b.reid_features) and torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor)
                ok &= torch.equal(a.scores, b.scores) and torch.equal(a.recs, b.recs) and torch.equal(a.bd, b.bd)
                ok &= torch.equal(a.ctrl_points, b.ctrl_points)
                ok &= np.array_equal(a._gom["boxes"], b.pred_boxes.tensor.numpy())
        digest = float(allrec.double().sum())
        q.put((rank, bool(ok), digest))
    finally:
        dist.destroy_process_group()


def test_allgather_records_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2], "ranks hold different gathered records"


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gomatching_amd.training import allreduce_gradients
        g = torch.Generator().manual_seed(7)
        params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in ((5, 3), (7,), (2, 2, 2))]
        frozen = torch.nn.Parameter(torch.randn(4, generator=g), requires_grad=False)   # frozen: left alone
        # gradients that exist on ONE rank only (a clip without ground-truth ids, a short-term matcher without rows): the
        # bucket must be the same on every rank, the missing gradient counting as zero
        only0 = torch.nn.Parameter(torch.randn(6, generator=g))
        nowhere = torch.nn.Parameter(torch.randn(3, generator=g))
        for i, p in enumerate(params):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        if rank == 0:
            only0.grad = torch.full_like(only0, 8.0)
        n = allreduce_gradients([params[0], frozen, only0] + params[1:] + [nowhere])
        ok = n == sum(p.numel() for p in params) + 6 + 3 and frozen.grad is None
        for i, p in enumerate(params):
            ok &= bool(torch.allclose(p.grad, torch.full_like(p, (1 + 2) / 2 * (i + 1))))
        ok &= only0.grad is not None and bool(torch.allclose(only0.grad, torch.full_like(only0, 4.0)))
        ok &= nowhere.grad is None                                  # unused on every rank: untouched, as under DDP
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_world2():
    """The data-parallel reduction of the head's gradients (gomatching_amd/training.py: one flattened bucket, averaged)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res), res


def _st_blocks(rank, frames=3):
    """The short-term score blocks a rank would hold for its `frames` frame pairs (indices rank * frames + j into the window):
    ragged sizes, an empty pair, a full nq x nq one."""
    g = np.random.default_rng(50 + rank)
    out = {}
    for j in range(frames):
        n_cur, n_prev = [(5, 7), (0, 0), (NQ, NQ)][j] if rank == 0 else [(1, NQ), (3, 2), (0, 0)][j]
        if n_cur and n_prev:
            out[rank * frames + j] = g.random((n_cur, n_prev), dtype=np.float32)
    return out


def _st_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = [rank * 3 + j for j in range(3)]
        blk = pack_short_term(_st_blocks(rank), mine, NQ)
        assert blk.shape == (3, 2 + NQ * NQ)
        allblk = all_gather_records(blk)
        got = unpack_short_term(allblk, list(range(3 * world)))
        want = {}
        for r in range(world):
            want.update(_st_blocks(r))
        ok = sorted(got) == sorted(want) and all(np.array_equal(got[t], want[t]) for t in want)
        q.put((rank, bool(ok), float(allblk.double().sum())))
    finally:
        dist.destroy_process_group()


def test_short_term_block_exchange_world2():
    """The SECOND exchange of a sharded step (dist.exchange_and_track): every rank scores the frame pairs it detected and the
    [F, 2 + nq^2] blocks are all-gathered; every rank must end up with every pair's matrix, bit for bit, in frame order."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_st_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]
