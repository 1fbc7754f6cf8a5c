"""Host side of training the association head -- FIRST PIECE ONLY: the integer ground-truth logic that
turns proposals + annotated instances into association targets.  The losses and their gradients are pinned as an oracle
(oracle/train_oracle.py, tests/golden/train_*.npz); the device side (forward with saved activations, HIP backward kernels
for FCHead4Query / the matcher transformers, RCCL all-reduce of the head's gradients) is not built yet, and
`GoMatching.forward` keeps raising NotImplementedError until it is.

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
