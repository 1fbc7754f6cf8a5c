"""CPU oracle: a plain-PyTorch fp32 restatement of the GoMatching inference hot path.

TEST INFRASTRUCTURE.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this file; the product (`gomatching_amd/`) never does.

Every function restates one row of SURVEY.md §8(a) and cites the reference lines it follows
(paths relative to /root/reference).  It is written functionally over a flat state-dict that
uses the reference's parameter names, and shares no code with the HIP host path.

Pinning status:
  * DeepSolo-without-backbone, matcher heads, tracker logic, detection/post-process: PINNED by
    `tests/golden/*.npz`, generated in the build container by running the reference's own
    modules (through oracle/ref_shim.py) on the same synthetic weights (oracle/gen_golden.py).
  * ResNet-50/FrozenBN backbone, Instances/Boxes/pairwise_iou/nms/ImageList, shortest-edge
    resize: these live in Detectron2 v0.6 / torchvision, which are NOT in /root/reference and not
    installed here -> restated from their published definitions, PARITY UNPINNED.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# A2  backbone: Detectron2 v0.6 ResNet-50, FrozenBN, STRIDE_IN_1X1=False  (external; unpinned)
#     call site: gomatching/modeling/meta_arch/gom_lstmatcher.py:42-61
# --------------------------------------------------------------------------
_STAGES = (("res2", 3, 1), ("res3", 4, 2), ("res4", 6, 2), ("res5", 3, 2))


def _conv_fbn(x, sd, name, stride=1, padding=0, relu=False, eps=1e-5):
    y = F.conv2d(x, sd[name + ".weight"], None, stride=stride, padding=padding)
    scale = sd[name + ".norm.weight"] * (sd[name + ".norm.running_var"] + eps).rsqrt()
    bias = sd[name + ".norm.bias"] - sd[name + ".norm.running_mean"] * scale
    y = y * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)
    return F.relu(y) if relu else y


def resnet50(x, sd, prefix="backbone.0.backbone."):
    """x: [B,3,H,W] normalised.  Returns {'res3','res4','res5'} NCHW."""
    x = _conv_fbn(x, sd, prefix + "stem.conv1", stride=2, padding=3, relu=True)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = {}
    for stage, nblk, first_stride in _STAGES:
        for i in range(nblk):
            p = "%s%s.%d." % (prefix, stage, i)
            s = first_stride if i == 0 else 1
            if (p + "shortcut.weight") in sd:
                sc = _conv_fbn(x, sd, p + "shortcut", stride=s)
            else:
                sc = x
            y = _conv_fbn(x, sd, p + "conv1", stride=1, relu=True)          # STRIDE_IN_1X1=False
            y = _conv_fbn(y, sd, p + "conv2", stride=s, padding=1, relu=True)
            y = _conv_fbn(y, sd, p + "conv3")
            x = F.relu(y + sc)
        outs[stage] = x
    return {k: outs[k] for k in ("res3", "res4", "res5")}


def mask_out_padding(feature_shapes, image_sizes, strides=(8, 16, 32)):
    """gom_lstmatcher.py:63-76."""
    masks = []
    for shape, st in zip(feature_shapes, strides):
        N, _, H, W = shape
        m = torch.ones((N, H, W), dtype=torch.bool)
        for i, (h, w) in enumerate(image_sizes):
            m[i, : int(np.ceil(float(h) / st)), : int(np.ceil(float(w) / st))] = 0
        masks.append(m)
    return masks


# --------------------------------------------------------------------------
# A3  PositionalEncoding2D  third_party/adet/layers/pos_encoding.py:62-82
# --------------------------------------------------------------------------
def pos_encoding_2d(mask, num_pos_feats=128, temperature=10000, scale=2 * math.pi):
    not_mask = ~mask
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    eps = 1e-6
    y_embed = (y_embed - 0.5) / (y_embed[:, -1:, :] + eps) * scale
    x_embed = (x_embed - 0.5) / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="trunc") / num_pos_feats)
    px = x_embed[:, :, :, None] / dim_t
    py = y_embed[:, :, :, None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


# --------------------------------------------------------------------------
# small helpers (adet/modeling/model/utils.py:7-37, adet/utils/misc.py:115-131)
# --------------------------------------------------------------------------
def linear(x, sd, name):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def layer_norm(x, sd, name):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def mlp(x, sd, name, n):
    for i in range(n):
        x = linear(x, sd, "%s.layers.%d" % (name, i))
        if i < n - 1:
            x = F.relu(x)
    return x


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def gen_point_pos_embed(pts, d_model, temp):
    scale = 2 * math.pi
    dim = d_model // 2
    dim_t = torch.arange(dim, dtype=torch.float32)
    dim_t = temp ** (2 * torch.div(dim_t, 2, rounding_mode="trunc") / dim)
    px = (pts[..., 0] * scale)[..., None] / dim_t
    py = (pts[..., 1] * scale)[..., None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=-1).flatten(-2)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=-1).flatten(-2)
    return torch.cat((px, py), dim=-1)


def mha(q_in, k_in, v_in, sd, name, nheads):
    """nn.MultiheadAttention forward, inputs [L,B,E] (seq-first), eval mode, no masks."""
    E = q_in.shape[-1]
    w, b = sd[name + ".in_proj_weight"], sd[name + ".in_proj_bias"]
    q = F.linear(q_in, w[:E], b[:E])
    k = F.linear(k_in, w[E:2 * E], b[E:2 * E])
    v = F.linear(v_in, w[2 * E:], b[2 * E:])
    Lq, B, _ = q.shape
    Lk = k.shape[0]
    hd = E // nheads
    q = q.reshape(Lq, B * nheads, hd).transpose(0, 1) * (1.0 / math.sqrt(hd))
    k = k.reshape(Lk, B * nheads, hd).transpose(0, 1)
    v = v.reshape(Lk, B * nheads, hd).transpose(0, 1)
    attn = torch.softmax(torch.bmm(q, k.transpose(1, 2)), dim=-1)
    o = torch.bmm(attn, v).transpose(0, 1).reshape(Lq, B, E)
    return linear(o, sd, name + ".out_proj")


# --------------------------------------------------------------------------
# A7  multi-scale deformable attention sampling
#     third_party/adet/layers/csrc/DeformAttn/ms_deform_im2col_cuda.cuh:33-84 (bilinear), :237-299 (kernel)
# --------------------------------------------------------------------------
def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, q_chunk=1024):
    """value [B,S,M,D]; spatial_shapes [L,2] (H,W) int64; sampling_loc [B,Lq,M,L,P,2] (x,y in [0,1]);
    attn_weight [B,Lq,M,L,P]  ->  [B,Lq,M*D].  Zero outside the map; per-corner bounds as the kernel.
    The queries are processed `q_chunk` at a time: every operation below is per (batch, query) row, so the chunking changes no
    bit of the result -- it keeps the ~20 temporaries of a level (each Lq x M x P x D floats: 245 MB at 60k tokens) in the CPU's
    caches, which is where a full-size frame's oracle time went (elementwise add / mul / compare over such temporaries)."""
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    if q_chunk and Lq > q_chunk:
        return torch.cat([ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc[:, q0:q0 + q_chunk],
                                                 attn_weight[:, q0:q0 + q_chunk], q_chunk=0) for q0 in range(0, Lq, q_chunk)], dim=1)
    out = value.new_zeros((B, Lq, M, D))
    vflat = value.reshape(B * S * M, D)
    bm = (torch.arange(B).view(B, 1, 1, 1) * S) * M + torch.arange(M).view(1, 1, M, 1)   # row of (b, s=0, m)
    for l in range(L):
        H, W = int(spatial_shapes[l, 0]), int(spatial_shapes[l, 1])
        start = int(level_start_index[l])
        loc = sampling_loc[:, :, :, l]                       # B,Lq,M,P,2
        w_im = loc[..., 0] * W - 0.5
        h_im = loc[..., 1] * H - 0.5
        inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
        h_low = torch.floor(h_im)
        w_low = torch.floor(w_im)
        lh, lw = h_im - h_low, w_im - w_low
        hh, hw = 1 - lh, 1 - lw
        h_low, w_low = h_low.long(), w_low.long()
        acc = value.new_zeros((B, Lq, M, P, D))
        for dy, dx, wt in ((0, 0, hh * hw), (0, 1, hh * lw), (1, 0, lh * hw), (1, 1, lh * lw)):
            yy, xx = h_low + dy, w_low + dx
            ok = inside & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
            idx = start + yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)
            v = vflat.index_select(0, (idx * M + bm).reshape(-1)).view(B, Lq, M, P, D)
            acc = acc + v * (wt * ok.to(value.dtype))[..., None]
        out = out + (acc * attn_weight[:, :, :, l, :, None]).sum(3)
    return out.reshape(B, Lq, M * D)


def msda_module(query, reference_points, input_flatten, spatial_shapes, level_start_index,
                padding_mask, sd, name, nheads, nlevels, npoints):
    """third_party/adet/layers/ms_deform_attn.py:117-156."""
    N, Lq, C = query.shape
    S = input_flatten.shape[1]
    value = linear(input_flatten, sd, name + ".value_proj")
    if padding_mask is not None:
        value = value.masked_fill(padding_mask[..., None], 0.0)
    value = value.view(N, S, nheads, C // nheads)
    off = linear(query, sd, name + ".sampling_offsets").view(N, Lq, nheads, nlevels, npoints, 2)
    aw = linear(query, sd, name + ".attention_weights").view(N, Lq, nheads, nlevels * npoints)
    aw = F.softmax(aw, -1).view(N, Lq, nheads, nlevels, npoints)
    normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
    loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    out = ms_deform_attn_forward(value, spatial_shapes, level_start_index, loc, aw)
    return linear(out, sd, name + ".output_proj")


# --------------------------------------------------------------------------
# A4-A10  DeepSolo without backbone
# --------------------------------------------------------------------------
def bernstein_matrix(num_points):
    """deformable_transformer.py:83-86 (float32 table, same arithmetic)."""
    from scipy.special import comb
    ts = torch.linspace(0, 1, num_points)
    rows = [[t ** k * (1 - t) ** (3 - k) * comb(3, k) for k in range(4)] for t in ts]
    return torch.tensor(rows)


def get_valid_ratio(mask):
    _, H, W = mask.shape
    vh = torch.sum(~mask[:, :, 0], 1).float() / H
    vw = torch.sum(~mask[:, 0, :], 1).float() / W
    return torch.stack([vw, vh], -1)


def encoder_reference_points(spatial_shapes, valid_ratios):
    """deformable_transformer.py:288-300."""
    lst = []
    for lvl, (H, W) in enumerate(spatial_shapes):
        ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing="ij")
        ry = ry.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
        rx = rx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
        lst.append(torch.stack((rx, ry), -1))
    ref = torch.cat(lst, 1)
    return ref[:, :, None] * valid_ratios[:, None]


def encoder_layer(src, pos, ref, shapes, lsi, mask, sd, p, T):
    """deformable_transformer.py:254-278."""
    src2 = msda_module(src + pos, ref, src, shapes, lsi, mask, sd, p + "self_attn",
                       T.NHEADS, T.NUM_FEATURE_LEVELS, T.DEC_N_POINTS)
    src = layer_norm(src + src2, sd, p + "norm1")
    src2 = linear(F.relu(linear(src, sd, p + "linear1")), sd, p + "linear2")
    return layer_norm(src + src2, sd, p + "norm2")


def decoder_layer(tgt, query_pos, ref_in, src, shapes, lsi, mask, sd, p, T):
    """deformable_transformer.py:372-427.  tgt/query_pos [B,nq,P,C]; ref_in [B,nq,P,L,2]."""
    B, nq, P, C = tgt.shape
    q = (tgt + query_pos).flatten(0, 1).transpose(0, 1)              # P, B*nq, C
    v = tgt.flatten(0, 1).transpose(0, 1)
    t2 = mha(q, q, v, sd, p + "attn_intra", T.NHEADS).transpose(0, 1).reshape(B, nq, P, C)
    tgt = layer_norm(tgt + t2, sd, p + "norm_intra")
    ti = tgt.transpose(1, 2)                                          # B,P,nq,C
    x = ti.flatten(0, 1).transpose(0, 1)                              # nq, B*P, C
    t2 = mha(x, x, x, sd, p + "attn_inter", T.NHEADS).transpose(0, 1).reshape(B, P, nq, C)
    ti = layer_norm(ti + t2, sd, p + "norm_inter").transpose(1, 2)    # B,nq,P,C
    t2 = msda_module((ti + query_pos).flatten(1, 2), ref_in.flatten(1, 2), src, shapes, lsi, mask,
                     sd, p + "attn_cross", T.NHEADS, T.NUM_FEATURE_LEVELS, T.ENC_N_POINTS).reshape(B, nq, P, C)
    tgt = layer_norm(ti + t2, sd, p + "norm_cross")
    t2 = linear(F.relu(linear(tgt, sd, p + "linear1")), sd, p + "linear2")
    return layer_norm(tgt + t2, sd, p + "norm3")


def deepsolo_forward(sd, cfg, feats, masks, pos, taps=None, prefix="detection_transformer.", topk_override=None):
    """detection_transformer_wobackbone.py:159-270 + deformable_transformer.py:150-215.
    feats: 3 NCHW tensors (res3/4/5); masks: 3 bool [B,H,W]; pos: 3 NCHW pos encodings."""
    T = cfg.MODEL.TRANSFORMER
    d, nq, P = T.HIDDEN_DIM, T.NUM_QUERIES, T.NUM_POINTS
    srcs, masks, pos = [], list(masks), list(pos)
    for l, f in enumerate(feats):
        y = F.conv2d(f, sd[prefix + "input_proj.%d.0.weight" % l], sd[prefix + "input_proj.%d.0.bias" % l])
        srcs.append(F.group_norm(y, 32, sd[prefix + "input_proj.%d.1.weight" % l],
                                 sd[prefix + "input_proj.%d.1.bias" % l], 1e-5))
    y = F.conv2d(feats[-1], sd[prefix + "input_proj.3.0.weight"], sd[prefix + "input_proj.3.0.bias"],
                 stride=2, padding=1)
    y = F.group_norm(y, 32, sd[prefix + "input_proj.3.1.weight"], sd[prefix + "input_proj.3.1.bias"], 1e-5)
    m = F.interpolate(masks[0][None].float(), size=y.shape[-2:]).to(torch.bool)[0]
    srcs.append(y)
    masks.append(m)
    pos.append(pos_encoding_2d(m, d // 2, T.TEMPERATURE))

    t = prefix + "transformer."
    shapes = [(s.shape[2], s.shape[3]) for s in srcs]
    src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
    mask = torch.cat([mm.flatten(1) for mm in masks], 1)
    lvl_pos = torch.cat([pp.flatten(2).transpose(1, 2) + sd[t + "level_embed"][l].view(1, 1, -1)
                         for l, pp in enumerate(pos)], 1)
    spatial_shapes = torch.as_tensor(shapes, dtype=torch.long)
    lsi = torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
    valid_ratios = torch.stack([get_valid_ratio(mm) for mm in masks], 1)
    if taps is not None:
        taps.update(src=src, lvl_pos=lvl_pos, spatial_shapes=spatial_shapes, level_start_index=lsi)

    # encoder
    ref = encoder_reference_points(shapes, valid_ratios)
    memory = src
    for i in range(T.ENC_LAYERS):
        memory = encoder_layer(memory, lvl_pos, ref, spatial_shapes, lsi, mask, sd,
                               t + "encoder.layers.%d." % i, T)
        if taps is not None and i == 0:
            taps["enc0"] = memory
    if taps is not None:
        taps["memory"] = memory

    # proposals (deformable_transformer.py:108-139, 183-199)
    B = memory.shape[0]
    props, cur = [], 0
    for (H, W) in shapes:
        mk = mask[:, cur:cur + H * W].view(B, H, W, 1)
        valid_H = torch.sum(~mk[:, :, 0, 0], 1)
        valid_W = torch.sum(~mk[:, 0, :, 0], 1)
        gy, gx = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij")
        grid = torch.cat([gx.unsqueeze(-1), gy.unsqueeze(-1)], -1)
        scale = torch.cat([valid_W.unsqueeze(-1), valid_H.unsqueeze(-1)], 1).view(B, 1, 1, 2)
        grid = (grid.unsqueeze(0).expand(B, -1, -1, -1) + 0.5) / scale
        props.append(grid.repeat(1, 1, 1, 4).view(B, -1, 8))
        cur += H * W
    props = torch.cat(props, 1)
    valid = ((props > 0.01) & (props < 0.99)).all(-1, keepdim=True)
    props = torch.log(props / (1 - props))
    props = props.masked_fill(mask.unsqueeze(-1), float("inf")).masked_fill(~valid, float("inf"))
    om = memory.masked_fill(mask.unsqueeze(-1), 0.0).masked_fill(~valid, 0.0)
    om = layer_norm(linear(om, sd, t + "enc_output"), sd, t + "enc_output_norm")
    enc_class = linear(om, sd, prefix + "bezier_proposal_class")
    enc_coord = mlp(om, sd, prefix + "bezier_proposal_coord", 3) + props
    topk = torch.topk(enc_class[..., 0], nq, dim=1)[1]
    if topk_override is not None:
        # test aid: the winners in a GIVEN rank order (a parity test hands over the HIP path's order where two winners' logits
        # are closer than any fp32 evaluation can resolve: a query slot = learned embedding + the token of that rank)
        topk = torch.as_tensor(topk_override, dtype=torch.long).view(B, nq)
    coords = torch.gather(enc_coord, 1, topk.unsqueeze(-1).repeat(1, 1, 8)).sigmoid()
    refp = torch.matmul(bernstein_matrix(P).to(coords.dtype), coords.view(B, nq, 4, 2))      # B,nq,P,2
    if taps is not None:
        taps.update(enc_class=enc_class[..., 0], topk=topk, init_ref=refp)

    # decoder (deformable_transformer.py:449-497)
    out = sd[prefix + "point_embed.weight"].reshape(nq, P, d).unsqueeze(0).expand(B, -1, -1, -1)
    refs = []
    for lid in range(T.DEC_LAYERS):
        ref_in = refp[:, :, :, None] * valid_ratios[:, None, None]
        qpos = mlp(gen_point_pos_embed(ref_in[:, :, :, 0, :], d, T.TEMPERATURE), sd,
                   t + "decoder.ref_point_head", 2)
        out = decoder_layer(out, qpos, ref_in, memory, spatial_shapes, lsi, mask, sd,
                            t + "decoder.layers.%d." % lid, T)
        tmp = mlp(out, sd, prefix + "ctrl_point_coord.0", 3)
        refp = (tmp + inverse_sigmoid(refp)).sigmoid()
        refs.append(refp)
        if taps is not None and lid == 0:
            taps["dec0"] = out
    hs = out

    # heads: last layer only, with inter_references[last-1]  (detection_transformer_wobackbone.py:209-253)
    reference = inverse_sigmoid(refs[T.DEC_LAYERS - 2])
    res = {
        "pred_logits": linear(hs, sd, prefix + "ctrl_point_class.0"),
        "pred_text_logits": linear(hs, sd, prefix + "ctrl_point_text.0"),
        "pred_ctrl_points": (mlp(hs, sd, prefix + "ctrl_point_coord.0", 3) + reference).sigmoid(),
        "pred_bd_points": None,
        "query_features": hs,
    }
    if T.BOUNDARY_HEAD:
        res["pred_bd_points"] = (mlp(hs, sd, prefix + "boundary_offset.0", 3) + reference.repeat(1, 1, 1, 2)).sigmoid()
    return res


# --------------------------------------------------------------------------
# Detectron2-like containers used by the tracker restatement (external; unpinned)
# --------------------------------------------------------------------------
class Inst(dict):
    """Minimal Instances: field dict + image_size; boolean/index selection over every field."""

    def __init__(self, image_size, **fields):
        super().__init__(**fields)
        self.image_size = image_size

    def __len__(self):
        for v in self.values():
            return len(v)
        return 0

    def select(self, idx):
        out = Inst(self.image_size)
        for k, v in self.items():
            out[k] = v[idx]
        return out


def pairwise_iou(a, b):
    area1 = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area2 = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = (torch.min(a[:, None, 2:], b[:, 2:]) - torch.max(a[:, None, :2], b[:, :2])).clamp(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (area1[:, None] + area2 - inter), torch.zeros(1))


def nms(boxes, scores, thr):
    """torchvision.ops.nms semantics: greedy, IoU > thr suppresses, output in decreasing-score order."""
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order].numpy().astype(np.float32)
    n = len(b)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(int(order[i]))
        r = b[i + 1:]
        if len(r) == 0:
            break
        w = np.maximum(np.float32(0), np.minimum(b[i, 2], r[:, 2]) - np.maximum(b[i, 0], r[:, 0]))
        h = np.maximum(np.float32(0), np.minimum(b[i, 3], r[:, 3]) - np.maximum(b[i, 1], r[:, 1]))
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (area[i] + area[i + 1:] - inter)
        dead[i + 1:] |= ovr > np.float32(thr)
    return torch.as_tensor(keep, dtype=torch.long)


# --------------------------------------------------------------------------
# A11-A13  detection(), NMS, FCHead4Query
# --------------------------------------------------------------------------
def detection(cfg, out, re_logits, image_sizes):
    """gom_lstmatcher.py:579-629 (boundary branch)."""
    thr = cfg.MODEL.TRANSFORMER.INFERENCE_TH_TEST
    text = torch.softmax(out["pred_text_logits"], dim=-1)
    scores = out["pred_logits"].mean(-2).sigmoid().max(-1)[0]
    if re_logits is not None:
        re = re_logits.mean(-2).sigmoid().max(-1)[0]
        scores = torch.where(scores > re, scores, re)
    results = []
    for b, image_size in enumerate(image_sizes):
        sel = scores[b] > thr
        cp = out["pred_ctrl_points"][b][sel].clone()
        cp[..., 0] *= image_size[1]
        cp[..., 1] *= image_size[0]
        bd = out["pred_bd_points"][b][sel].clone()
        bd[..., 0::2] *= image_size[1]
        bd[..., 1::2] *= image_size[0]
        n = int(sel.sum())
        results.append(Inst(
            image_size,
            scores=scores[b][sel],
            pred_classes=torch.zeros(n, dtype=torch.long),
            ctrl_points=cp.flatten(1),
            recs=text[b][sel].topk(1)[1].squeeze(-1),
            bd=bd,
            query_features=out["query_features"][b][sel],
        ))
    return results


def proposals_with_nms(cfg, det):
    """gom_lstmatcher.py:310-332."""
    out = []
    for r in det:
        if len(r) > 0:
            pts = r["bd"].reshape(len(r), -1, 2)
            boxes = torch.cat([pts[:, :, 0].min(-1)[0][:, None], pts[:, :, 1].min(-1)[0][:, None],
                               pts[:, :, 0].max(-1)[0][:, None], pts[:, :, 1].max(-1)[0][:, None]], -1)
            keep = nms(boxes, r["scores"], cfg.VIDEO_TEST.NMS_THRESH)
            p = r.select(keep)
            p["proposal_boxes"] = boxes[keep]
        else:
            p = r.select(slice(None))
            p["proposal_boxes"] = torch.zeros((0, 4))
        p["objectness_logits"] = p["scores"]
        out.append(p)
    return out


def roi_heads_forward(sd, cfg, proposals, prefix="roi_heads."):
    """lstmatcher.py:271-290,546-557 (eval) + association_head.py:116-122."""
    name = cfg.MODEL.ROI_HEADS.NAME
    thr = cfg.MODEL.TRANSFORMER.INFERENCE_TH_TEST if name == "SHA_FFN_CRSATTN" \
        else cfg.MODEL.ASSO_HEAD.ASSO_THRESH_TEST
    if thr <= 0:
        thr = cfg.MODEL.ASSO_HEAD.ASSO_THRESH
    res = []
    for p in proposals:
        q = p.select(p["objectness_logits"] > thr)
        x = q["query_features"].flatten(1)
        for k in range(cfg.MODEL.ASSO_HEAD.NUM_FC):
            x = F.relu(linear(x, sd, prefix + "asso_head.fc%d" % (k + 1)))
        r = Inst(q.image_size, reid_features=x, pred_boxes=q["proposal_boxes"], scores=q["objectness_logits"],
                 pred_classes=torch.zeros(len(q), dtype=torch.long), ctrl_points=q["ctrl_points"],
                 recs=q["recs"], bd=q["bd"])
        res.append(r)
    return res


# --------------------------------------------------------------------------
# A14-A15  matcher transformers + association scores
# --------------------------------------------------------------------------
def matcher_transformer(sd, cfg, reid, query_inds, short_term, prefix="roi_heads."):
    """roi_heads/transformer.py:60-96 with the shipped switches (norm=False, no decoder self-attn).
    reid [N,F] -> (feat [M,F], memory [N,F])."""
    A = cfg.MODEL.ASSO_HEAD
    H = A.NUM_HEADS
    src = reid[:, None, :]                                             # N,1,F  (seq-first, B=1)
    if cfg.MODEL.ROI_HEADS.NAME == "SHA_FFN_CRSATTN":
        name, n_enc, only_crs = prefix + "shared_matcher", 0, True
    else:
        name = prefix + ("short_term_matcher" if short_term else "long_term_matcher")
        n_enc, only_crs = A.NUM_ENCODER_LAYERS, False
    memory = src
    for i in range(n_enc):
        p = "%s.encoder.layers.%d." % (name, i)
        memory = memory + mha(memory, memory, memory, sd, p + "self_attn", H)
        memory = memory + linear(F.relu(linear(memory, sd, p + "linear1")), sd, p + "linear2")
    tgt = src[query_inds] if query_inds is not None else src
    for i in range(A.NUM_DECODER_LAYERS):
        p = "%s.decoder.layers.%d." % (name, i)
        tgt = tgt + mha(tgt, memory, memory, sd, p + "multihead_attn", H)
        if not only_crs:
            tgt = tgt + linear(F.relu(linear(tgt, sd, p + "linear1")), sd, p + "linear2")
    return tgt[:, 0], memory[:, 0]


def asso_scores(sd, cfg, reid, n_t, k, short_term):
    """lstmatcher.py:333-381: transformer -> q.mem^T -> per-frame softmax with a zero 'background' logit."""
    lo, hi = sum(n_t[:k]), sum(n_t[:k + 1])
    feat, mem = matcher_transformer(sd, cfg, reid, list(range(lo, hi)), short_term)
    asso = feat @ mem.t()                                              # n_k x N
    outs = []
    for a in asso.split(n_t, dim=1):
        outs.append(torch.cat([a, a.new_zeros((a.shape[0], 1))], dim=1).softmax(dim=1)[:, :-1])
    return torch.cat(outs, dim=1)


def _norm_boxes(instances):
    bs = []
    for p in instances:
        h, w = p.image_size
        b = p["pred_boxes"].clone()
        b[:, [0, 2]] /= w
        b[:, [1, 3]] /= h
        bs.append(b)
    return torch.cat(bs, 0)


def _lsa(cost):
    from scipy.optimize import linear_sum_assignment
    return linear_sum_assignment(cost.numpy())


class MatchLog:
    """Margins of the tracker's DISCRETE decisions, and a way to let a near-tie fall the other way (test infrastructure).

    A match (gom_lstmatcher.py:434-452 / :521-554) makes two kinds of decisions on its `traj` matrix: the linear-sum
    assignment on -traj, and `traj[i, j] > thr_j` for every assigned pair.  For every match `decide` records
      * `assign_gap`: total score of the optimal assignment minus that of the best assignment with a DIFFERENT outcome (found by
        forbidding each optimal pair in turn and solving again: the second-best assignment lacks at least one of them);
      * `thr_margin`: the smallest |traj[i, j] - thr_j| over the optimal pairs.
    With `eps` > 0 the alternatives whose gap / margin is below eps are enumerated (`n_alt`), and `script` = {call index: k} makes
    call number `index` take its k-th alternative (k >= 1) instead of the optimum: tests replay the clip with the HIP path's
    decision forced at a near-tie and require every later id to follow (tests/helpers.py track_clip_tie_aware)."""

    def __init__(self, eps=0.0, script=None):
        self.eps = float(eps)
        self.script = dict(script or {})
        self.calls = []
        self.seconds = 0.0                                             # time spent in here (bench.py takes it out of the CPU baseline)

    @staticmethod
    def _outcome(traj, thr, mi, mj, flip=None):
        out = [-1] * traj.shape[0]
        for i, j in zip(mi, mj):
            ok = bool(traj[i, j] > thr[j])
            if flip is not None and (int(i), int(j)) == flip:
                ok = not ok
            if ok:
                out[int(i)] = int(j)
        return tuple(out)

    def decide(self, kind, frame, traj, thr, mi, mj):
        """-> (mi, mj, flip): the assignment to use and the one pair whose threshold test is inverted (or None)."""
        import time
        t0 = time.time()
        try:
            return self._decide(kind, frame, traj, thr, mi, mj)
        finally:
            self.seconds += time.time() - t0

    def _decide(self, kind, frame, traj, thr, mi, mj):
        best = self._outcome(traj, thr, mi, mj)
        total = float(traj[mi, mj].sum()) if len(mi) else 0.0
        alts = {}                                                      # outcome -> (gap, kind, payload)
        gap_min = float("inf")
        for i, j in zip(mi, mj):
            if float(traj[i, j]) == 0.0 and float(thr[j]) >= 0.0:
                continue                                               # a zero pair is never accepted: swapping it changes nothing
            t2 = traj.clone()
            t2[i, j] = -1e6
            ai, aj = _lsa(-t2)
            if any(int(a) == int(i) and int(b) == int(j) for a, b in zip(ai, aj)):
                continue                                               # (forced back onto the pair: no alternative without it)
            out = self._outcome(traj, thr, ai, aj)
            if out == best:
                continue
            gap = total - float(traj[ai, aj].sum())
            gap_min = min(gap_min, gap)
            if gap < self.eps and (out not in alts or gap < alts[out][0]):
                alts[out] = (gap, "assign", (ai, aj))
        thr_min = float("inf")
        for i, j in zip(mi, mj):
            m = abs(float(traj[i, j]) - float(thr[j]))
            thr_min = min(thr_min, m)
            if m < self.eps:
                out = self._outcome(traj, thr, mi, mj, flip=(int(i), int(j)))
                if out not in alts or m < alts[out][0]:
                    alts[out] = (m, "flip", (int(i), int(j)))
        ordered = sorted(alts.values(), key=lambda a: a[0])
        idx = len(self.calls)
        pick = int(self.script.get(idx, 0))
        if pick > len(ordered):
            pick = 0
        self.calls.append({"frame": frame, "kind": kind, "rows": int(traj.shape[0]), "cols": int(traj.shape[1]),
                           "assign_gap": gap_min, "thr_margin": thr_min, "n_alt": len(ordered), "picked": pick,
                           "picked_gap": ordered[pick - 1][0] if pick else 0.0})
        if pick == 0:
            return mi, mj, None
        _, what, payload = ordered[pick - 1]
        if what == "assign":
            return payload[0], payload[1], None
        return mi, mj, payload

    def summary(self):
        """Smallest margins over the clip (what a reader needs to judge how close the ids sit to a flip).  `exact_ties`: matches whose
        best alternative has EXACTLY the optimum's total (scores that are the same float -- two detections whose association softmax
        saturates at exactly 1.0 for one track, or IoU = 1.0: the tie then falls by the assignment solver's order, SciPy's in both
        implementations);
        `min_nonzero_assign_gap`: the smallest gap among the others."""
        fin = lambda v: None if v == float("inf") else v
        gaps = [c["assign_gap"] for c in self.calls]
        return {"matches": len(self.calls), "exact_ties": sum(1 for g_ in gaps if g_ == 0.0),
                "min_assign_gap": fin(min(gaps, default=float("inf"))),
                "min_nonzero_assign_gap": fin(min([g_ for g_ in gaps if g_ > 0.0], default=float("inf"))),
                "min_thr_margin": fin(min([c["thr_margin"] for c in self.calls], default=float("inf")))}


def _thr_vector(V, id_inds):
    if V.NOT_MULT_THRESH:
        return torch.full((id_inds.shape[1],), float(V.OVERLAP_THRESH))
    return V.OVERLAP_THRESH * id_inds.sum(dim=0)


def run_short_term_match(sd, cfg, instances, id_count=None, log=None, frame=None):
    """gom_lstmatcher.py:405-465.  instances = [prev, cur]; sets cur['track_ids']."""
    V = cfg.VIDEO_TEST
    n_t = [len(x) for x in instances]
    N = sum(n_t)
    reid = torch.cat([x["reid_features"] for x in instances], 0)
    asso = asso_scores(sd, cfg, reid, n_t, 1, True)
    boxes = _norm_boxes(instances)
    n_k = n_t[1]
    ids = instances[0]["track_ids"].view(-1)
    Np = N - n_k
    k_inds = list(range(n_t[0], N))
    nonk = [i for i in range(N) if i not in k_inds]
    asso_nonk = asso[:, nonk]
    k_boxes, nonk_boxes = boxes[k_inds], boxes[nonk]
    uniq = torch.unique(ids)
    id_inds = (uniq[None, :] == ids[:, None]).float()
    traj = torch.mm(asso_nonk, id_inds)
    if id_inds.numel() > 0:
        last = (id_inds * torch.arange(Np)[:, None]).max(dim=0)[1]
        ious = pairwise_iou(k_boxes, nonk_boxes[last])
    else:
        ious = traj.new_zeros(traj.shape)
    if V.WITH_IOU:
        traj = torch.max(traj, ious)
    mi, mj = _lsa(-traj)
    flip = None
    if log is not None:
        mi, mj, flip = log.decide("short", frame, traj, _thr_vector(V, id_inds), mi, mj)
    track_ids = ids.new_full((n_k,), -1)
    for i, j in zip(mi, mj):
        thr = V.OVERLAP_THRESH * id_inds[:, j].sum() if not V.NOT_MULT_THRESH else V.OVERLAP_THRESH
        if bool(traj[i, j] > thr) != (flip == (int(i), int(j))):
            track_ids[i] = uniq[j]
    if id_count:
        for i in range(n_k):
            if track_ids[i] < 0:
                id_count = id_count + 1
                track_ids[i] = id_count
    instances[1]["track_ids"] = track_ids
    if id_count:
        return instances, id_count
    return instances, torch.unique(track_ids)


def run_long_term_match(sd, cfg, full, k, id_count, cur_id, log=None, frame=None):
    """gom_lstmatcher.py:467-564."""
    V = cfg.VIDEO_TEST
    cur = set(int(c) for c in cur_id)
    insts, reid_idx = [], None
    for idx, p in enumerate(full):
        if idx != len(full) - 1:
            keep = torch.tensor([int(t) not in cur for t in p["track_ids"]], dtype=torch.bool)
            q = Inst(full[0].image_size, track_ids=p["track_ids"][keep])
        else:
            keep = torch.tensor([int(t) == -1 for t in p["track_ids"]], dtype=torch.bool)
            reid_idx = keep
            q = Inst(full[0].image_size)
        q["reid_features"] = p["reid_features"][keep]
        q["pred_boxes"] = p["pred_boxes"][keep]
        insts.append(q)
    n_t = [len(x) for x in insts]
    N, T = sum(n_t), len(n_t)
    reid = torch.cat([x["reid_features"] for x in insts], 0)
    asso = asso_scores(sd, cfg, reid, n_t, k, False)
    boxes = _norm_boxes(insts)
    n_k = n_t[k]
    Np = N - n_k
    ids = torch.cat([x["track_ids"] for t, x in enumerate(insts) if t != k], 0).view(Np)
    k_inds = list(range(sum(n_t[:k]), sum(n_t[:k + 1])))
    nonk = [i for i in range(N) if i not in k_inds]
    asso_nonk = asso[:, nonk]
    k_boxes, nonk_boxes = boxes[k_inds], boxes[nonk]
    uniq = torch.unique(ids)
    id_inds = (uniq[None, :] == ids[:, None]).float()
    if V.DECAY_TIME > 0:
        dts = torch.cat([reid.new_full((len(x),), T - t - 2) for t, x in enumerate(insts) if t != k], 0)
        asso_nonk = asso_nonk * (V.DECAY_TIME ** dts[None, :])
    traj = torch.mm(asso_nonk, id_inds)
    if id_inds.numel() > 0:
        last = (id_inds * torch.arange(Np)[:, None]).max(dim=0)[1]
        ious = pairwise_iou(k_boxes, nonk_boxes[last])
    else:
        ious = traj.new_zeros(traj.shape)
    if V.WITH_IOU:
        traj = torch.max(traj, ious)
    if V.MAX_CENTER_DIST > 0.0:
        k_ct = (k_boxes[:, :2] + k_boxes[:, 2:]) / 2
        k_s = ((k_boxes[:, 2:] - k_boxes[:, :2]) ** 2).sum(dim=1)
        n_ct = (nonk_boxes[:, :2] + nonk_boxes[:, 2:]) / 2
        dist = ((k_ct[:, None] - n_ct[None, :]) ** 2).sum(dim=2)
        valid = dist / (k_s[:, None] + 1e-8) < V.MAX_CENTER_DIST
        valid_assn = torch.mm(valid.float(), id_inds).clamp_(max=1.0).long().bool()
        traj[~valid_assn] = 0
    mi, mj = _lsa(-traj)
    flip = None
    if log is not None:
        mi, mj, flip = log.decide("long", frame, traj, _thr_vector(V, id_inds), mi, mj)
    track_ids = ids.new_full((n_k,), -1)
    for i, j in zip(mi, mj):
        thr = V.OVERLAP_THRESH * id_inds[:, j].sum() if not V.NOT_MULT_THRESH else V.OVERLAP_THRESH
        if bool(traj[i, j] > thr) != (flip == (int(i), int(j))):
            track_ids[i] = uniq[j]
    for i in range(n_k):
        if track_ids[i] < 0:
            id_count = id_count + 1
            track_ids[i] = id_count
    full[k]["track_ids"][reid_idx] = track_ids
    return full, id_count


def track_clip(sd, cfg, per_frame, batch_id=0, id_count=0, instances=None, log=None):
    """The control flow of gom_lstmatcher.py:366-403 over already-detected frames
    (`per_frame` = output of roi_heads_forward, one Inst per frame).  `log`: a MatchLog (margins / forced near-ties)."""
    test_len = cfg.INPUT.VIDEO.TEST_LEN
    instances = [] if instances is None else instances
    start = batch_id * 100
    for f, inst in enumerate(per_frame):
        instances.append(inst)
        rf = start + f
        if rf == 0:
            instances[0]["track_ids"] = torch.arange(1, len(instances[0]) + 1)
            id_count = len(instances[0]) + 1
        elif rf == 1:
            pair, id_count = run_short_term_match(sd, cfg, instances[rf - 1: rf + 1], id_count=id_count, log=log, frame=rf)
            instances[rf - 1: rf + 1] = pair
        else:
            pair, cur_id = run_short_term_match(sd, cfg, instances[rf - 1: rf + 1], log=log, frame=rf)
            instances[rf - 1: rf + 1] = pair
            if -1 in cur_id:
                st, ed = max(0, rf + 1 - test_len), rf + 1
                win, id_count = run_long_term_match(sd, cfg, instances[st:ed], min(test_len - 1, rf),
                                                    id_count, cur_id, log=log, frame=rf)
                instances[st:ed] = win
        assert len(instances[-1]["track_ids"]) == len(torch.unique(instances[-1]["track_ids"]))
        if rf - test_len >= 0:
            instances[rf - test_len].pop("reid_features", None)
    return instances, id_count


def remove_short_track(cfg, instances):
    """gom_lstmatcher.py:566-577."""
    ids = torch.cat([x["track_ids"] for x in instances], 0)
    uniq = ids.unique()
    id_inds = (uniq[:, None] == ids[None, :]).float()
    uniq = uniq.clone()
    uniq[id_inds.sum(dim=1) < cfg.VIDEO_TEST.MIN_TRACK_LEN] = -1
    ids = uniq[torch.where(id_inds.permute(1, 0))[1]].split([len(x) for x in instances])
    return [inst.select(ids[k] >= 0) for k, inst in enumerate(instances)]


def batch_postprocess(instances, image_sizes, min_size=None, max_size=None):
    """gom_lstmatcher.py:353-364 + :78-111: scale ctrl_points and bd, not pred_boxes.  min_size / max_size are only set
    for the ViTAE backbone (:144-146): the scale then comes from the resize rule, not from the (padded) tensor shape."""
    out = []
    for r, (height, width) in zip(instances, image_sizes):
        if min_size and max_size:
            size = min_size * 1.0
            k = min_size / min(width, height)
            newh, neww = (size, k * width) if height < width else (k * height, size)
            if max(newh, neww) > max_size:
                k = max_size * 1.0 / max(newh, neww)
                newh, neww = newh * k, neww * k
            sx, sy = width / int(neww + 0.5), height / int(newh + 0.5)
        else:
            sx, sy = width / r.image_size[1], height / r.image_size[0]
        r["ctrl_points"][:, 0::2] *= sx
        r["ctrl_points"][:, 1::2] *= sy
        r["bd"][..., 0::2] *= sx
        r["bd"][..., 1::2] *= sy
        out.append({"instances": r})
    return out


# --------------------------------------------------------------------------
# A1 + whole-frame driver
# --------------------------------------------------------------------------
def detect_frames(sd, cfg, images, taps=None, topk_override=None):
    """gom_lstmatcher.py:268-351 for a list of [3,H,W] float images (0..255, cfg.INPUT.FORMAT order),
    all of one size: returns the per-frame Inst list *before* tracking."""
    mean = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1)
    std = torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1)
    x = torch.stack([(im - mean) / std for im in images])
    sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in images]
    if cfg.MODEL.BACKBONE.NAME == "build_swin_backbone":
        from oracle.swin_oracle import swin_tiny
        feats = swin_tiny(x, sd)
        feats = [feats[k] for k in ("stage3", "stage4", "stage5")]
    elif cfg.MODEL.BACKBONE.NAME == "build_vitaev2_backbone":
        from oracle.vitae_oracle import vitae_v2_s
        feats = vitae_v2_s(x, sd)
        feats = [feats[k] for k in ("stage3", "stage4", "stage5")]
    else:
        feats = resnet50(x, sd)
        feats = [feats[k] for k in ("res3", "res4", "res5")]
    masks = mask_out_padding([f.shape for f in feats], sizes)
    T = cfg.MODEL.TRANSFORMER
    pos = [pos_encoding_2d(m, T.HIDDEN_DIM // 2, T.TEMPERATURE) for m in masks]
    if taps is not None:
        taps.update(res3=feats[0], res4=feats[1], res5=feats[2])
    out = deepsolo_forward(sd, cfg, feats, masks, pos, taps=taps, topk_override=topk_override)
    re = linear(out["query_features"], sd, "roi_heads.rescoring_head") if cfg.MODEL.ROI_HEADS.WITH_RESR else None
    if taps is not None:
        taps.update({("out_" + k): v for k, v in out.items() if v is not None})
        if re is not None:
            taps["re_logits"] = re
    det = detection(cfg, out, re, sizes)
    props = proposals_with_nms(cfg, det)
    return roi_heads_forward(sd, cfg, props)


def run_clip(sd, cfg, images, orig_hw=None, log=None, per_frame=None):
    """Whole path for one clip (GoMBatchPredictor.__call__ window, text_track_visualizer.py:325-334).  per_frame: the frames'
    detections when a caller already has them (`detect_frames` of each image on its own: tests share them between cases)."""
    with torch.no_grad():
        if per_frame is None:
            per_frame = []
            for im in images:                               # one frame at a time, as the reference does
                per_frame.extend(detect_frames(sd, cfg, [im]))
        assert len(per_frame) == len(images)
        instances, id_count = track_clip(sd, cfg, per_frame, log=log)
        if cfg.VIDEO_TEST.MIN_TRACK_LEN > 0:
            instances = remove_short_track(cfg, instances)
        if orig_hw is None:
            sizes = [(im.shape[-2], im.shape[-1]) for im in images]
        elif isinstance(orig_hw[0], (tuple, list)):             # one source size per frame (mixed-resolution clip)
            sizes = [tuple(s) for s in orig_hw]
        else:
            sizes = [tuple(orig_hw)] * len(instances)
        vitae = cfg.MODEL.BACKBONE.NAME == "build_vitaev2_backbone"
        return batch_postprocess(instances, sizes, cfg.INPUT.MIN_SIZE_TEST if vitae else None,
                                 cfg.INPUT.MAX_SIZE_TEST if vitae else None), id_count
