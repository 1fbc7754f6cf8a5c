"""Training of the association head on MI355X (gom_lstmatcher.py:213-266, lstmatcher.py:237-330,384-475).

Two halves.  Host: the integer ground-truth logic that turns proposals + annotated instances into association targets and the
Hungarian matching of `loss_res` (numpy + the library's LSA).  Device (second half of this file): the training forward of
the trainable head with saved activations, its backward on HIP kernels, the loss dict of `GoMatching.forward`, and the
gradient all-reduce.  Losses AND gradients are pinned against the reference's own (`tests/golden/train_*.npz`).

`association_targets` mirrors `LSTMatcher._get_asso_gt` (lstmatcher.py:388-433; identical in shared_ffn_crsattn.py) on
numpy arrays: it is bookkeeping over at most a few hundred boxes per clip and stays on the host, as the id logic of the
tracker does.
"""
import numpy as np


def normalised_boxes_and_times(boxes_per_frame, image_sizes):
    """`_get_boxes_time` (lstmatcher.py:478-496): boxes [n_t,4] px per frame -> ([N,4] float32 in image units, [N] frame index)."""
    out, times = [], []
    for t, (b, (h, w)) in enumerate(zip(boxes_per_frame, image_sizes)):
        b = np.array(b, dtype=np.float32).reshape(-1, 4).copy()
        b[:, [0, 2]] /= np.float32(w)
        b[:, [1, 3]] /= np.float32(h)
        out.append(b)
        times.append(np.full((b.shape[0],), t, np.int64))
    return (np.concatenate(out) if out else np.zeros((0, 4), np.float32)), \
        (np.concatenate(times) if times else np.zeros((0,), np.int64))


def pairwise_iou(a, b):
    """detectron2.structures.pairwise_iou in float32: inter / (area_a + area_b - inter), 0 where the boxes do not overlap."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = np.clip(np.minimum(a[:, None, 2:], b[None, :, 2:]) - np.maximum(a[:, None, :2], b[None, :, :2]), 0, None)
    inter = wh[..., 0] * wh[..., 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / (area_a[:, None] + area_b[None, :] - inter)
    return np.where(inter > 0, iou, np.float32(0)).astype(np.float32)


def association_targets(pred_box, pred_time, target_box, target_time, target_inst_id, n_t):
    """(gt [K,T] int64, match_cues [N] int64).  gt[k, t] = index (within frame t) of the proposal overlapping track k's
    annotated box in frame t, or n_t[t] ("background") when the track is absent or unmatched there; match_cues[j] = the
    track proposal j belongs to, or -1.  Tracks are the sorted distinct positive instance ids."""
    pred_time, target_time = np.asarray(pred_time), np.asarray(target_time)
    target_inst_id = np.asarray(target_inst_id, np.int64)
    ious = pairwise_iou(pred_box, target_box)
    ious[pred_time[:, None] != target_time[None, :]] = -1.0
    inst_ids = np.unique(target_inst_id[target_inst_id > 0])
    K, N, T = len(inst_ids), len(pred_time), len(n_t)
    cues = np.full((N,), -1, np.int64)
    gt = np.zeros((K, T), np.int64)
    starts = np.concatenate([[0], np.cumsum(n_t)]).astype(np.int64)
    for k, inst in enumerate(inst_ids):
        sel = target_inst_id == inst
        for t in range(T):
            block = ious[starts[t]:starts[t + 1]][:, sel]
            if block.size == 0:
                gt[k, t] = n_t[t]
                continue
            val, ind = block.max(axis=0), block.argmax(axis=0)      # per annotated box of the track: its best proposal
            hit = ind[val > 0.0]
            if len(hit) > 1:
                raise ValueError("track %d has %d overlapping proposals' maxima in frame %d (the reference asserts <= 1)"
                                 % (int(inst), len(hit), t))
            if len(hit) == 1:
                gt[k, t] = int(hit[0])
                cues[starts[t] + int(hit[0])] = k
            else:
                gt[k, t] = n_t[t]
    return gt, cues


def point_matching(re_logits, pred_ctrl_points, target_ctrl_points, focal_alpha=0.25, focal_gamma=2.0, class_weight=1.0,
                   coord_weight=1.0):
    """`CtrlPointHungarianMatcher4GM.forward` (third_party/adet/modeling/model/matcher.py:175-198) for ONE image on numpy
    arrays: re_logits [nq, P, 1] (rescoring head), pred_ctrl_points [nq, P, 2], target_ctrl_points [g, P, 2] (normalised) ->
    (query indices [min(nq,g)], target indices) of the minimum-cost assignment, solved by the library's host LSA
    (`gom_linear_sum_assignment`, SciPy's tie-breaking)."""
    from . import ops
    x = np.asarray(re_logits, np.float32)
    prob = (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)
    neg = (1 - focal_alpha) * (prob ** focal_gamma) * (-np.log(1 - prob + np.float32(1e-8)))
    pos = focal_alpha * ((1 - prob) ** focal_gamma) * (-np.log(prob + np.float32(1e-8)))
    cost_class = (pos[..., 0] - neg[..., 0]).mean(-1, keepdims=True)                    # [nq, 1]
    a = np.asarray(pred_ctrl_points, np.float32).reshape(x.shape[0], -1)
    b = np.asarray(target_ctrl_points, np.float32).reshape(-1, a.shape[1])
    cost_pts = np.abs(a[:, None, :] - b[None, :, :]).sum(-1)                             # torch.cdist(p=1)
    cost = class_weight * cost_class + coord_weight * cost_pts
    return ops.linear_sum_assignment(cost.astype(np.float64))


# =====================================================================================================================
# Device side: the training forward / backward of the trainable head on HIP kernels.
#
# Only `roi_heads` trains in the reference (gomatching/modeling/freeze_layers.py:20-37; train_net.py:111-131): FCHead4Query,
# the matcher transformer(s) and the rescoring head.  Every contraction of the forward AND of the backward (dgrad = dY W,
# wgrad = dY^T X, bias grad = dY^T 1) runs on the library's exact-fp32 MFMA GEMM (`gom_gemm_f32` / its split-K form) with
# transposed operands made by `gom_transpose_f32`; attention is the same GEMM per head around `gom_softmax_rows_scaled_f32`
# and its backward kernel; the losses are `gom_asso_ce_f32` / `gom_sigmoid_focal_f32` (csrc/train.hip).  torch supplies
# device memory, the autograd tape that strings the kernels together, and the glue around them (concatenation, zero-padded
# copies, the arithmetic on the 0-d loss values).
# Pinned by the reference's own losses AND gradients (tests/golden/train_asso_*.npz, train_res_ic15.npz;
# tests/test_training_gpu.py).
# =====================================================================================================================
def _torch():
    import torch
    return torch


def _pad4(n):
    return (n + 3) // 4 * 4


def _transpose_padded(x):
    """x [R, C] (row-strided) -> x^T as [C, pad4(R)] with zero columns beyond R (the GEMM wants K % 4 == 0)."""
    torch = _torch()
    from . import ops
    R, C = x.shape
    out = torch.zeros((C, _pad4(R)), dtype=torch.float32, device=x.device)
    if R and C:
        ops.transpose_into(x, out)
    return out


def _make_functions():
    """autograd Functions are created lazily so that importing this module never needs torch or the GPU library."""
    torch = _torch()
    from . import ops

    class Linear(torch.autograd.Function):
        """y = act(x W^T + b); x [M, K] row-strided, W [N, K]."""

        @staticmethod
        def forward(ctx, x, w, b, relu):
            y = ops.gemm(x, w, bias=b, relu=bool(relu))
            ctx.save_for_backward(x, w, y if relu else None)
            ctx.relu, ctx.has_b = bool(relu), b is not None
            return y

        @staticmethod
        def backward(ctx, dy):
            x, w, y = ctx.saved_tensors
            dy = dy.contiguous()
            if ctx.relu:
                dy = ops.relu_backward(dy, y)
            M, N = dy.shape
            dx = dw = db = None
            if M == 0:
                return torch.zeros_like(x), torch.zeros_like(w), (torch.zeros((N,), device=w.device) if ctx.has_b else None), None
            if ctx.needs_input_grad[0]:
                dyp = dy
                if N % 4:                                                                             # the GEMM wants K % 4 == 0
                    dyp = torch.zeros((M, _pad4(N)), dtype=torch.float32, device=dy.device)
                    dyp[:, :N] = dy
                dx = ops.gemm(dyp, _transpose_padded(w))                                              # dY W   (K = pad4(N))
            dyT = _transpose_padded(dy)                                                               # [N, Mp]
            if ctx.needs_input_grad[1]:
                dw = ops.gemm(dyT, _transpose_padded(x))                                              # dY^T X (K = Mp)
            if ctx.has_b and ctx.needs_input_grad[2]:
                ones = torch.zeros((1, dyT.shape[1]), dtype=torch.float32, device=dy.device)
                ones[:, :M] = 1.0
                db = ops.gemm(dyT, ones).view(N)
            return dx, dw, db, None

    class Attention(torch.autograd.Function):
        """softmax(q k^T / sqrt(hd)) v per head for ONE sequence pair: q [Lq, E], k, v [Lk, E] (nn.MultiheadAttention core,
        transformer.py:208,287 of the reference's matcher), heads laid side by side in E."""

        @staticmethod
        def forward(ctx, q, k, v, heads):
            Lq, E = q.shape
            Lk = k.shape[0]
            hd = E // heads
            Lkp = _pad4(Lk)
            out = torch.zeros((Lq, E), dtype=torch.float32, device=q.device)
            P = torch.zeros((heads, Lq, Lkp), dtype=torch.float32, device=q.device)
            scale = 1.0 / math.sqrt(hd)
            if Lq and Lk:
                for h in range(heads):
                    c = slice(h * hd, (h + 1) * hd)
                    ops.gemm(q[:, c], k[:, c], out=P[h][:, :Lk])                                       # q_h k_h^T
                    ops.softmax_rows_scaled_(P[h], Lk, scale)
                    ops.gemm(P[h], _transpose_padded(v[:, c]), out=out[:, c])                         # P v_h   (K = Lkp)
            ctx.save_for_backward(q, k, v, P)
            ctx.heads, ctx.scale = heads, scale
            return out

        @staticmethod
        def backward(ctx, do):
            q, k, v, P = ctx.saved_tensors
            heads, scale = ctx.heads, ctx.scale
            Lq, E = q.shape
            Lk = k.shape[0]
            hd = E // heads
            dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
            if Lq == 0 or Lk == 0:
                return dq, dk, dv, None
            do = do.contiguous()
            for h in range(heads):
                c = slice(h * hd, (h + 1) * hd)
                Ph = P[h]
                ops.gemm(_transpose_padded(Ph[:, :Lk]), _transpose_padded(do[:, c]), out=dv[:, c])    # P^T dO  (K = Lqp)
                dP = torch.zeros_like(Ph)
                ops.gemm(do[:, c], v[:, c], out=dP[:, :Lk])                                           # dO v_h^T
                dS = ops.softmax_rows_backward(Ph, dP, Lk, scale)
                ops.gemm(dS, _transpose_padded(k[:, c]), out=dq[:, c])                                # dS k_h   (K = Lkp)
                ops.gemm(_transpose_padded(dS[:, :Lk]), _transpose_padded(q[:, c]), out=dk[:, c])     # dS^T q_h (K = Lqp)
            return dq, dk, dv, None

    class Add(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a, b):
            return ops.add(a.contiguous(), b.contiguous())

        @staticmethod
        def backward(ctx, g):
            return g, g

    class AssoCE(torch.autograd.Function):
        """sum over (row, frame) of the per-frame cross entropy with background (detr_asso_loss, lstmatcher.py:436-475)."""

        @staticmethod
        def forward(ctx, logits, offs, gt):
            loss, _ = ops.asso_ce(logits, offs, gt)
            ctx.save_for_backward(logits, offs, gt)
            # the sum of a handful of values: one GEMM against a ones row (K padded to 4)
            flat = torch.zeros((1, _pad4(loss.numel())), dtype=torch.float32, device=logits.device)
            flat[0, :loss.numel()] = loss.view(-1)
            return ops.gemm(flat, torch.ones_like(flat)).view(())

        @staticmethod
        def backward(ctx, g):
            logits, offs, gt = ctx.saved_tensors
            _, dl = ops.asso_ce(logits, offs, gt, want_loss=False, grad_scale=g.reshape(1).contiguous().float())
            return dl, None, None

    class FocalSum(torch.autograd.Function):
        """sum of the sigmoid focal loss over every element (loss_res, lstmatcher.py:237-268)."""

        @staticmethod
        def forward(ctx, x, target, alpha, gamma):
            loss, dx = ops.sigmoid_focal(x, target, alpha, gamma)
            ctx.save_for_backward(dx)
            flat = torch.zeros((1, _pad4(loss.numel())), dtype=torch.float32, device=x.device)
            flat[0, :loss.numel()] = loss.view(-1)
            return ops.gemm(flat, torch.ones_like(flat)).view(())

        @staticmethod
        def backward(ctx, g):
            (dx,) = ctx.saved_tensors
            return dx * g, None, None, None

    return {"Linear": Linear, "Attention": Attention, "Add": Add, "AssoCE": AssoCE, "FocalSum": FocalSum}


_FN = None


def _fn():
    global _FN
    if _FN is None:
        _FN = _make_functions()
    return _FN


import math  # noqa: E402


def _linear(x, params, name, relu=False):
    return _fn()["Linear"].apply(x, params[name + ".weight"], params.get(name + ".bias"), relu)


def _mha(q_in, kv_in, params, name, heads):
    """nn.MultiheadAttention(q, k = v = kv_in) with packed in_proj (transformer.py:208,287; eval-mode dropout)."""
    E = q_in.shape[1]
    w, b = params[name + ".in_proj_weight"], params[name + ".in_proj_bias"]
    L = _fn()["Linear"]
    q = L.apply(q_in, w[:E], b[:E], False)
    kv = L.apply(kv_in, w[E:], b[E:], False)
    a = _fn()["Attention"].apply(q, kv[:, :E], kv[:, E:], heads)
    return L.apply(a, params[name + ".out_proj.weight"], params[name + ".out_proj.bias"], False)


def matcher_transformer(params, cfg, reid, short_term, prefix="roi_heads."):
    """Training forward of the matcher (roi_heads/transformer.py:60-96 with norm = Identity, every proposal a query):
    returns (feats [N, F], memory [N, F])."""
    A = cfg.MODEL.ASSO_HEAD
    shared = cfg.MODEL.ROI_HEADS.NAME == "SHA_FFN_CRSATTN"
    name = prefix + ("shared_matcher" if shared else ("short_term_matcher" if short_term else "long_term_matcher"))
    add = _fn()["Add"].apply
    memory = reid
    n_enc = 0 if shared else A.NUM_ENCODER_LAYERS
    for i in range(n_enc):
        p = "%s.encoder.layers.%d." % (name, i)
        memory = add(memory, _mha(memory, memory, params, p + "self_attn", A.NUM_HEADS))
        h = _linear(memory, params, p + "linear1", relu=True)
        memory = add(memory, _linear(h, params, p + "linear2"))
    tgt = reid
    for i in range(A.NUM_DECODER_LAYERS):
        p = "%s.decoder.layers.%d." % (name, i)
        tgt = add(tgt, _mha(tgt, memory, params, p + "multihead_attn", A.NUM_HEADS))
        if not shared:
            h = _linear(tgt, params, p + "linear1", relu=True)
            tgt = add(tgt, _linear(h, params, p + "linear2"))
    return tgt, memory


def _detr_asso_loss(logits, gt, cues, n_t, neg_unmatched):
    """detr_asso_loss (lstmatcher.py:436-475) on the device: logits [N, N]; gt [K, T] / cues [N] from association_targets."""
    torch = _torch()
    N, T = logits.shape[0], len(n_t)
    tgt = np.full((N, T), -1, np.int32)
    src = np.nonzero(cues >= 0)[0]
    if neg_unmatched:
        tgt[:] = np.asarray(n_t, np.int32)[None, :]
    if len(src):
        tgt[src] = gt[cues[src]].astype(np.int32)
    counted = tgt >= 0
    num = float(((tgt != np.asarray(n_t, np.int32)[None, :]) & counted).sum())
    offs = np.concatenate([[0], np.cumsum(n_t)]).astype(np.int32)
    dev = logits.device
    total = _fn()["AssoCE"].apply(logits.contiguous(), torch.from_numpy(offs).to(dev), torch.from_numpy(tgt).to(dev))
    return total / (num + 1e-4)


def asso_losses(params, cfg, frames, targets, prefix="roi_heads."):
    """`_forward_asso`, training branch (lstmatcher.py:271-330 = shared_ffn_crsattn.py), on the device.
    params: {state-dict key: CUDA tensor (nn.Parameter)}; frames: per frame {"image_size", "proposal_boxes" [n,4] px,
    "objectness_logits" [n], "query_features" [n,25,256]} (CUDA); targets: per frame {"image_size", "gt_boxes" [g,4] px,
    "gt_instance_ids" [g]} (host or device).  Returns {"loss_long_asso", "loss_short_asso"} with autograd history."""
    torch = _torch()
    A = cfg.MODEL.ASSO_HEAD
    dev = params[prefix + "asso_head.fc1.weight"].device
    keep = [(f["objectness_logits"] > A.ASSO_THRESH).nonzero().flatten() for f in frames]
    boxes = [f["proposal_boxes"][k].detach().cpu().numpy() for f, k in zip(frames, keep)]
    n_t = [int(len(k)) for k in keep]
    sizes = [tuple(f["image_size"]) for f in frames]
    width = int(np.prod(frames[0]["query_features"].shape[1:]))                 # 25 points x 256 channels
    qf = [f["query_features"][k].reshape(len(k), width) for f, k in zip(frames, keep)]
    x = torch.cat(qf).to(dev).float().contiguous()
    for i in range(A.NUM_FC):
        x = _linear(x, params, prefix + "asso_head.fc%d" % (i + 1), relu=True)
    reid = x
    tb = [np.asarray(t["gt_boxes"].cpu() if hasattr(t["gt_boxes"], "cpu") else t["gt_boxes"], np.float32).reshape(-1, 4)
          for t in targets]
    tids = [np.asarray(t["gt_instance_ids"].cpu() if hasattr(t["gt_instance_ids"], "cpu") else t["gt_instance_ids"], np.int64)
            for t in targets]
    zero = reid.sum() * 0.0 if reid.numel() else torch.zeros((), device=dev)
    if sum(len(b) for b in tb) == 0 or max(int(i.max()) for i in tids if len(i)) == 0:
        return {"loss_long_asso": zero, "loss_short_asso": zero}

    def one(lo_f, hi_f, reid_sl, short):
        feats, memory = matcher_transformer(params, cfg, reid_sl, short, prefix)
        logits = _fn()["Linear"].apply(feats, memory, None, False)                 # ATTWeightHead, 0 layers: q . k^T
        pb, pt = normalised_boxes_and_times(boxes[lo_f:hi_f], sizes[lo_f:hi_f])
        gb, gtime = normalised_boxes_and_times(tb[lo_f:hi_f], [tuple(t["image_size"]) for t in targets[lo_f:hi_f]])
        ids = np.concatenate([i for i in tids[lo_f:hi_f] if len(i)]) if any(len(i) for i in tids[lo_f:hi_f]) else np.zeros((0,), np.int64)
        gt, cues = association_targets(pb, pt, gb, gtime, ids, n_t[lo_f:hi_f])
        return _detr_asso_loss(logits, gt, cues, n_t[lo_f:hi_f], A.NEG_UNMATCHED)

    loss_long = one(0, len(frames), reid, False)
    loss_short, eff = zero, 0
    for c in range(1, len(frames)):
        ids_sl = [i for i in tids[c - 1:c + 1] if len(i)]
        if not ids_sl or max(int(i.max()) for i in ids_sl) == 0:
            continue
        eff += 1
        lo, hi = sum(n_t[:c - 1]), sum(n_t[:c + 1])
        loss_short = loss_short + one(c - 1, c + 1, reid[lo:hi], True)
    loss_short = loss_short / (eff + 1e-4)
    return {"loss_long_asso": A.ASSO_WEIGHT * loss_long, "loss_short_asso": A.ASSO_WEIGHT_LOCAL * loss_short}


def loss_res(params, cfg, query_features, pred_ctrl_points, targets, prefix="roi_heads."):
    """`LSTMatcher.loss_res` (lstmatcher.py:237-268): rescoring head (Linear 256 -> 1 on every point feature) under the sigmoid
    focal loss against the Hungarian-matched queries.  query_features [B,nq,P,256], pred_ctrl_points [B,nq,P,2] (frozen detector,
    CUDA); targets: per image {"labels" [g], "ctrl_points" [g,P,2]}."""
    torch = _torch()
    Lc = cfg.MODEL.TRANSFORMER.LOSS
    B, nq, P, C = query_features.shape
    logits = _linear(query_features.reshape(-1, C).float().contiguous(), params, prefix + "rescoring_head")     # [B*nq*P, 1]
    lg_host = logits.detach().view(B, nq, P, 1).cpu().numpy()
    pts_host = pred_ctrl_points.detach().cpu().numpy()
    onehot = np.zeros((B, nq, P, 1), np.float32)
    num_inst = 0
    for b, t in enumerate(targets):
        tc = np.asarray(t["ctrl_points"].cpu() if hasattr(t["ctrl_points"], "cpu") else t["ctrl_points"], np.float32)
        num_inst += len(tc)
        if len(tc) == 0:
            continue
        src, _ = point_matching(lg_host[b], pts_host[b], tc, Lc.FOCAL_ALPHA, Lc.FOCAL_GAMMA, Lc.POINT_CLASS_WEIGHT,
                                Lc.POINT_COORD_WEIGHT)
        onehot[b, src] = 1.0                                      # one text class: label 0 of NUM_CLASSES = 1
    target = torch.from_numpy(onehot.reshape(-1, 1)).to(logits.device)
    total = _fn()["FocalSum"].apply(logits, target, float(Lc.FOCAL_ALPHA), float(Lc.FOCAL_GAMMA))
    return {"loss_res": total / (P * max(float(num_inst), 1.0))}


def allreduce_gradients(parameters, group=None):
    """Data-parallel training of the head (train_net.py runs under DistributedDataParallel): ONE all-reduce (RCCL over xGMI with
    backend "nccl"; gloo in the CPU test) of all gradients flattened into one bucket, averaged over the ranks.  The head has
    12-33 M parameters (47-131 MB): one bucket is latency-optimal on the point-to-point xGMI ring.
    The bucket is RANK-INVARIANT: every parameter with `requires_grad`, in the order given, zeros where this rank has no
    gradient (a clip without ground-truth ids returns the zero loss, a clip whose short-term matcher had no rows leaves that
    matcher's parameters without one -- on that rank only); the averaged gradient is written back to every parameter that had a
    gradient on AT LEAST ONE rank, as DistributedDataParallel does.  A parameter no rank used keeps `grad = None` (one
    "has a gradient" word per parameter travels at the end of the bucket), so weight decay / momentum treat it exactly as a
    single-process run does.  No rank ever skips the collective."""
    torch = _torch()
    import torch.distributed as dist
    params = [p for p in parameters if p.requires_grad]
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    if not params:
        return 0                                                 # rank-invariant: `requires_grad` is a property of the model
    has = torch.tensor([0.0 if p.grad is None else 1.0 for p in params], dtype=params[0].dtype, device=params[0].device)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params] + [has])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    used = flat[-len(params):].cpu() > 0
    flat = flat[:-len(params)]
    flat /= dist.get_world_size(group)
    o = 0
    for p, u in zip(params, used.tolist()):
        n = p.numel()
        if not u:
            pass                                                 # unused on every rank: stays None
        elif p.grad is None:
            p.grad = flat[o:o + n].view_as(p).clone()
        else:
            p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n
    return flat.numel()


def forward_losses(model, batched_inputs):
    """`GoMatching.forward` in training (gom_lstmatcher.py:213-266) for the META_ARCH wrapper (`compat/d2_register.py`): the
    frozen detector runs on the HIP inference kernels without a tape; the losses of the trainable head carry autograd
    history onto `model.roi_heads`' parameters.  batched_inputs: the reference's list of {"image" [3,H,W], "instances": ground
    truth with gt_boxes / gt_instance_ids / (for loss_res) normalised ctrl points under `polyline` or `ctrl_points`}."""
    torch = _torch()
    from . import ops
    from .predictor import new_time_cost
    if hasattr(model, "impl"):                                   # the META_ARCH wrapper: live nn.Parameters
        impl = model.impl(for_training=True)
        params = {k: p for k, p in model.named_parameters() if k.startswith("roi_heads.")}
    else:                                                        # the HIP model itself: its own trainable copies
        impl = model
        params = model.trainable_parameters()
    cfg = model.cfg
    T = cfg.MODEL.TRANSFORMER
    B, nq, P = len(batched_inputs), T.NUM_QUERIES, T.NUM_POINTS
    with torch.no_grad():
        tc = new_time_cost()
        raw, kind = impl._raw_input(batched_inputs)
        x = impl._normalise(raw, kind)
        feats = impl.backbone.forward(x)
        out = impl.detection_transformer.forward([feats[k] for k in impl.feature_names])
        qf = out["query_features"].view(B, nq, P, -1)
        re = None
        if impl.with_rescore:
            re = ops.gemm(out["query_features"], params["roi_heads.rescoring_head.weight"].detach(),
                          bias=params["roi_heads.rescoring_head.bias"].detach())
        recs = ops.argmax_rows(out["pred_text_logits"])
        # training proposals: no NMS (gom_lstmatcher.py:231-258 builds them straight from `detection`), kept by the score threshold
        det = ops.detect_post(out["pred_logits"], re, out["pred_ctrl_points"], out["pred_bd_points"], recs, B, nq, P,
                              kind[1][0], kind[1][1], impl.test_score_threshold, 2.0, -1.0)
        torch.cuda.current_stream().synchronize()
        if ops.GEMM_MODE == "f16x3":
            # the range flag of the f16x3 kernels (`detect_launch` reads and clears it per inference step): an activation beyond
            # fp16's range during a TRAINING forward must not feed the losses silently, nor stay set for the next inference step
            ops.check_range_flag(qf.device)
    counts = det["count"].cpu().numpy()
    keep = det["keep_idx"].cpu().numpy()
    frames, targets, res_targets = [], [], []
    for b, inp in enumerate(batched_inputs):
        n = int(counts[b])
        rows = torch.from_numpy(keep[b, :n].astype(np.int64)).to(qf.device)      # rows of the flattened [B*nq] query axis
        frames.append({"image_size": kind[1], "proposal_boxes": det["boxes"][b, :n], "objectness_logits": det["scores"][b, :n],
                       "query_features": qf.view(B * nq, P, -1).index_select(0, rows)})
        gt = inp["instances"]
        get = (lambda k: gt.get(k)) if hasattr(gt, "get") else (lambda k: gt[k])
        boxes = get("gt_boxes")
        boxes = boxes.tensor if hasattr(boxes, "tensor") else boxes
        targets.append({"image_size": kind[1], "gt_boxes": boxes, "gt_instance_ids": get("gt_instance_ids")})
        if impl.with_rescore:
            pts = get("ctrl_points") if (hasattr(gt, "has") and gt.has("ctrl_points")) or (isinstance(gt, dict) and "ctrl_points" in gt) \
                else get("polyline")
            pts = torch.as_tensor(pts).float().reshape(-1, P, 2) / torch.tensor([kind[1][1], kind[1][0]], dtype=torch.float32)
            res_targets.append({"labels": np.zeros((pts.shape[0],), np.int64), "ctrl_points": pts})
    losses = asso_losses(params, cfg, frames, targets)
    if impl.with_rescore:
        losses.update(loss_res(params, cfg, qf, out["pred_ctrl_points"].view(B, nq, P, 2), res_targets))
    return losses
