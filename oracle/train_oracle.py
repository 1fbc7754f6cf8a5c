"""CPU restatement of the TRAINING forward of the association head (lstmatcher.py:271-330, 373-475 and the
identical code of shared_ffn_crsattn.py): from per-frame proposals (boxes, objectness, DeepSolo query features) and
ground-truth instances to {'loss_long_asso', 'loss_short_asso'}.  TEST INFRASTRUCTURE: imported only by tests/ and the
fixture generator.  Plain differentiable torch: autograd through it gives the head's gradients.

PINNED by tests/golden/train_asso_*.npz, which oracle/gen_golden_train.py produces by calling the reference's own
`_forward_asso` in training mode (dropout switched off: it is the one stochastic element) on the repo's synthetic weights.
Only `roi_heads` trains in the reference (freeze_layers.py:20-37): the association losses above and `loss_res` of the
rescoring head (further down; pinned by tests/golden/train_res_ic15.npz).
"""
import torch
import torch.nn.functional as F

from oracle import gom_oracle as O


def _pairwise_iou(a, b):
    """detectron2.structures.pairwise_iou: inter / (area_a + area_b - inter), 0 where inter is 0."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = (torch.min(a[:, None, 2:], b[None, :, 2:]) - torch.max(a[:, None, :2], b[None, :, :2])).clamp(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (area_a[:, None] + area_b[None, :] - inter), torch.zeros_like(inter))


def _boxes_time(frames, key):
    """_get_boxes_time (:478-496): boxes normalised by the frame size, frame index per box."""
    boxes, times = [], []
    for t, p in enumerate(frames):
        h, w = p["image_size"]
        b = p[key].clone()
        b[:, [0, 2]] /= w
        b[:, [1, 3]] /= h
        boxes.append(b)
        times.append(torch.full((b.shape[0],), t, dtype=torch.long))
    return torch.cat(boxes).detach(), torch.cat(times).detach()


def asso_gt(pred_box, pred_time, target_box, target_time, target_inst_id, n_t):
    """_get_asso_gt (:388-433): per (track k, frame t) the index of the proposal that overlaps the track's ground truth
    box in that frame (n_t[t] = 'background'), and per proposal the track it belongs to (-1: none)."""
    ious = _pairwise_iou(pred_box, target_box)
    ious[pred_time[:, None] != target_time[None, :]] = -1.0
    inst_ids = torch.unique(target_inst_id[target_inst_id > 0])
    K, N, T = len(inst_ids), len(pred_box), len(n_t)
    match_cues = torch.full((N,), -1, dtype=torch.long)
    ret = torch.zeros((K, T), dtype=torch.long)
    per_frame = ious.split(n_t, dim=0)
    for k, inst_id in enumerate(inst_ids):
        sel = target_inst_id == inst_id
        base = 0
        for t in range(T):
            iou_t = per_frame[t][:, sel]
            if iou_t.numel() == 0:
                ret[k, t] = n_t[t]
            else:
                val, inds = iou_t.max(dim=0)
                ind = inds[val > 0.0]
                assert len(ind) <= 1
                if len(ind) == 1:
                    ret[k, t] = int(ind[0])
                    match_cues[base + int(ind[0])] = k
                else:
                    ret[k, t] = n_t[t]
            base += n_t[t]
    return ret, match_cues


def detr_asso_loss(asso_pred, gt, match_cues, n_t, neg_unmatched):
    """detr_asso_loss + _match (:436-475): per frame a cross entropy over [proposals of the frame | background]."""
    src = torch.where(match_cues >= 0)[0]
    tgt = match_cues[src]
    loss, num = 0, 0
    zero = asso_pred.new_zeros((asso_pred.shape[0], 1))
    for t, a in enumerate(asso_pred.split(n_t, dim=1)):
        logits = torch.cat([a, zero], dim=1)
        if neg_unmatched:
            gt_t = torch.full((asso_pred.shape[0],), n_t[t], dtype=torch.long)
            gt_t[src] = gt[tgt, t]
        else:
            logits = logits[src]
            gt_t = gt[tgt, t]
        num = num + (gt_t != n_t[t]).float().sum()
        loss = loss + F.cross_entropy(logits, gt_t, reduction="none")
    return loss.sum() / (num + 1e-4)


def asso_losses(sd, cfg, frames, targets):
    """_forward_asso, training branch.  frames: list of {"image_size", "proposal_boxes" [n,4] px, "objectness_logits" [n],
    "query_features" [n,25,256]}; targets: list of {"image_size", "gt_boxes" [g,4] px, "gt_instance_ids" [g]}."""
    A = cfg.MODEL.ASSO_HEAD
    keep = [p["objectness_logits"] > A.ASSO_THRESH for p in frames]
    props = [{"image_size": p["image_size"], "proposal_boxes": p["proposal_boxes"][k], "query_features": p["query_features"][k]}
             for p, k in zip(frames, keep)]
    x = torch.cat([p["query_features"] for p in props]).flatten(1)
    for i in range(A.NUM_FC):
        x = torch.relu(O.linear(x, sd, "roi_heads.asso_head.fc%d" % (i + 1)))
    reid = x
    n_t = [len(p["proposal_boxes"]) for p in props]
    zero = reid.new_zeros((1,))[0]
    if sum(len(t["gt_boxes"]) for t in targets) == 0 or \
            max(int(t["gt_instance_ids"].max()) for t in targets if len(t["gt_boxes"]) > 0) == 0:
        return {"loss_long_asso": zero, "loss_short_asso": zero}

    def one(frames_sl, targets_sl, reid_sl, n_sl, short):
        feat, mem = O.matcher_transformer(sd, cfg, reid_sl, None, short)          # every proposal is a query (M = N)
        logits = feat @ mem.t()
        pb, pt = _boxes_time(frames_sl, "proposal_boxes")
        tb, tt = _boxes_time(targets_sl, "gt_boxes")
        ids = torch.cat([t["gt_instance_ids"] for t in targets_sl if len(t["gt_boxes"]) > 0])
        gt, cues = asso_gt(pb, pt, tb, tt, ids, n_sl)
        return detr_asso_loss(logits, gt, cues, n_sl, A.NEG_UNMATCHED)

    loss_long = one(props, targets, reid, n_t, False)
    loss_short, eff = 0, 0
    for c in range(1, len(props)):
        tsl = targets[c - 1:c + 1]
        if sum(len(t["gt_boxes"]) for t in tsl) == 0 or max(int(t["gt_instance_ids"].max()) for t in tsl if len(t["gt_boxes"]) > 0) == 0:
            continue
        eff += 1
        lo, hi = sum(n_t[:c - 1]), sum(n_t[:c + 1])
        loss_short = loss_short + one(props[c - 1:c + 1], tsl, reid[lo:hi], n_t[c - 1:c + 1], True)
    loss_short = loss_short / (eff + 1e-4)
    return {"loss_long_asso": A.ASSO_WEIGHT * loss_long, "loss_short_asso": A.ASSO_WEIGHT_LOCAL * loss_short}


def point_matching(cfg, re_logits, pred_ctrl_points, targets):
    """CtrlPointHungarianMatcher4GM.forward (third_party/adet/modeling/model/matcher.py:175-198): focal class cost of the
    RESCORED logits averaged over the 25 points + L1 distance of the control points, one assignment per image."""
    from scipy.optimize import linear_sum_assignment
    L = cfg.MODEL.TRANSFORMER.LOSS
    with torch.no_grad():
        sizes = [len(t["labels"]) for t in targets]
        bs, nq = re_logits.shape[:2]
        prob = re_logits.flatten(0, 1).sigmoid()
        out_pts = pred_ctrl_points.flatten(0, 1).flatten(-2)
        tgt_pts = torch.cat([t["ctrl_points"] for t in targets]).flatten(-2)
        neg = (1 - L.FOCAL_ALPHA) * (prob ** L.FOCAL_GAMMA) * (-(1 - prob + 1e-8).log())
        pos = L.FOCAL_ALPHA * ((1 - prob) ** L.FOCAL_GAMMA) * (-(prob + 1e-8).log())
        cost_class = (pos[..., 0] - neg[..., 0]).mean(-1, keepdims=True)
        C = L.POINT_CLASS_WEIGHT * cost_class + L.POINT_COORD_WEIGHT * torch.cdist(out_pts, tgt_pts, p=1)
        C = C.view(bs, nq, -1)
        out = []
        for i, c in enumerate(C.split(sizes, -1)):
            r, col = linear_sum_assignment(c[i].numpy())
            out.append((torch.as_tensor(r, dtype=torch.int64), torch.as_tensor(col, dtype=torch.int64)))
        return out


def loss_res(sd, cfg, query_features, pred_ctrl_points, targets):
    """LSTMatcher.loss_res (lstmatcher.py:237-268) for one process: the rescoring head (Linear 256 -> 1 on every point
    feature, :185-186) under a sigmoid focal loss against the Hungarian-matched queries.  query_features [B,nq,25,256] and
    pred_ctrl_points [B,nq,25,2] are the frozen detector's outputs; targets: {"labels" [g], "ctrl_points" [g,25,2]}."""
    L = cfg.MODEL.TRANSFORMER.LOSS
    num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    logits = O.linear(query_features, sd, "roi_heads.rescoring_head")            # [B,nq,25,1]
    indices = point_matching(cfg, logits, pred_ctrl_points, targets)
    num_inst = max(float(sum(len(t["labels"]) for t in targets)), 1.0)
    batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
    src_idx = torch.cat([src for src, _ in indices])
    target_classes = torch.full(logits.shape[:-1], num_classes, dtype=torch.int64)
    matched = torch.cat([t["labels"][j] for t, (_, j) in zip(targets, indices)])
    target_classes[batch_idx, src_idx] = matched[..., None]
    onehot = torch.zeros(list(logits.shape[:-1]) + [logits.shape[-1] + 1], dtype=logits.dtype)
    onehot.scatter_(-1, target_classes.unsqueeze(-1), 1)
    onehot = onehot[..., :-1]
    prob = logits.sigmoid()
    ce = F.binary_cross_entropy_with_logits(logits, onehot, reduction="none")
    p_t = prob * onehot + (1 - prob) * (1 - onehot)
    loss = ce * ((1 - p_t) ** L.FOCAL_GAMMA)
    if L.FOCAL_ALPHA >= 0:
        loss = (L.FOCAL_ALPHA * onehot + (1 - L.FOCAL_ALPHA) * (1 - onehot)) * loss
    return {"loss_res": loss.mean((1, 2)).sum() / num_inst * logits.shape[1]}
