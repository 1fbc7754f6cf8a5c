"""CPU restatement of the ViTAEv2-S backbone of the reference (third_party/adet/modeling/vitae_v2/*.py, built by
`build_vitaev2_backbone`, vitae_v2.py:228-249).  TEST INFRASTRUCTURE: imported only by tests/ and the fixture generator.

Functional torch-CPU code over the reference's own state-dict key names (prefix `backbone.0.backbone.`), eval mode
(DropPath / Dropout identities, BatchNorm2d on its running statistics).  PINNED by tests/golden/vitae_s.npz, which
oracle/gen_golden_vitae.py produces by running the reference's unmodified `ViTAEv2` module on the repo's synthetic weights.

Input sizes must be multiples of 32: at any other size the reference's own ReductionCell fails (`assert N == H * W`,
ReductionCell.py:143, or the `.view(*x.shape)` of the conv branch two lines below it).
"""
import math

import torch
import torch.nn.functional as F

EMBED, TOKEN = (64, 64, 128, 256), (64, 128, 256, 512)
RATIOS, KERNEL = (4, 2, 2, 2), (7, 3, 3, 3)
DILATIONS = ((1, 2, 3, 4), (1, 2, 3), (1, 2), (1, 2))
RC_HEADS, NC_HEADS = (1, 1, 2, 4), (1, 2, 4, 8)
RC_GROUP, NC_GROUP = (1, 16, 32, 64), (1, 32, 64, 128)
NC_DEPTH = (2, 2, 8, 2)
TYPES = ("window", "window", "transformer", "transformer")
WINDOW = 7
PREFIX = "backbone.0.backbone."


def _ln(x, sd, name, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def _lin(x, sd, name):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _bn(x, sd, name):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"],
                        False, 0.0, 1e-5)


def _pcm(x, sd, p, strides, groups):
    """nn.Sequential(conv, BN, SiLU, conv, BN, SiLU, conv) (ReductionCell.py:97-105, NormalCell.py:137-145)."""
    x = F.conv2d(x, sd[p + "0.weight"], sd[p + "0.bias"], strides[0], 1, 1, groups)
    x = F.silu(_bn(x, sd, p + "1"))
    x = F.conv2d(x, sd[p + "3.weight"], sd[p + "3.bias"], strides[1], 1, 1, groups)
    x = F.silu(_bn(x, sd, p + "4"))
    return F.conv2d(x, sd[p + "6.weight"], sd[p + "6.bias"], strides[2], 1, 1, groups)


def _mlp(x, sd, p):
    return _lin(F.gelu(_lin(x, sd, p + "fc1")), sd, p + "fc2")


def _window_attention(xn, H, W, sd, p, heads, out_dim):
    """Centred zero padding to multiples of 7, window partition, WindowAttention.forward (window.py:92-124; no relative
    position bias, no shift in this model), window reverse, crop (ReductionCell.py:145-163 / NormalCell.py:160-211)."""
    B, _, C = xn.shape
    ws = WINDOW
    td, lr = (ws - H % ws) % ws, (ws - W % ws) % ws
    top, left = td // 2, lr // 2
    x = F.pad(xn.view(B, H, W, C).permute(0, 3, 1, 2), (left, lr - left, top, td - top)).permute(0, 2, 3, 1)
    Hp, Wp = H + td, W + lr
    xw = x.reshape(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    B_, N = xw.shape[0], ws * ws
    qkv = _lin(xw, sd, p + "qkv").reshape(B_, N, 3, heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (out_dim // heads) ** -0.5, qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)).softmax(-1)
    y = _lin((attn @ v).transpose(1, 2).reshape(B_, N, -1), sd, p + "proj")
    y = y.view(B, Hp // ws, Wp // ws, ws, ws, out_dim).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, out_dim)
    return y[:, top:top + H, left:left + W, :].reshape(B, H * W, out_dim)


def _token_attention(xn, sd, p, heads, in_dim):
    """token_transformer.py:27-44: full attention whose skip connection is V (input and output widths differ)."""
    B, N, _ = xn.shape
    qkv = _lin(xn, sd, p + "qkv").reshape(B, N, 3, heads, in_dim // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = ((q @ k.transpose(-2, -1)) * (in_dim // heads) ** -0.5).softmax(-1)
    x = _lin((attn @ v).transpose(1, 2).reshape(B, N, in_dim), sd, p + "proj")
    return v.permute(0, 2, 1, 3).reshape(B, N, in_dim) + x


def _full_attention(xn, sd, p, heads):
    """NormalCell.Attention.forward (NormalCell.py:46-58)."""
    B, N, C = xn.shape
    qkv = _lin(xn, sd, p + "qkv").reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = ((q @ k.transpose(-2, -1)) * (C // heads) ** -0.5).softmax(-1)
    return _lin((attn @ v).transpose(1, 2).reshape(B, N, C), sd, p + "proj")


def reduction_cell(x, sd, i):
    """ReductionCell.forward (ReductionCell.py:133-188) on an NCHW map; returns tokens [B, h*w, T] and (h, w)."""
    p = PREFIX + "layers.%d.RC." % i
    B, _, H, W = x.shape
    r, k = RATIOS[i], KERNEL[i]
    ys = []
    for j, d in enumerate(DILATIONS[i]):                     # PRM (ReductionCell.py:27-34, 55-62)
        pad = math.ceil(((k - 1) * d + 1 - r) / 2)
        ys.append(F.gelu(F.conv2d(x, sd[p + "PRM.convs.%d.0.weight" % j], sd[p + "PRM.convs.%d.0.bias" % j], r, pad, d)))
    h, w = H // r, W // r
    assert all(tuple(y.shape[-2:]) == (h, w) for y in ys), "the reference asserts here too (ReductionCell.py:143)"
    prm = torch.cat(ys, 1).flatten(2).permute(0, 2, 1)       # channel = dilation * E + c ('cat', :63-64)
    strides, res = [], r // 2
    for _ in range(3):
        strides.append((res > 0) + 1)
        res //= 2
    conv = _pcm(x, sd, p + "PCM.", strides, RC_GROUP[i]).permute(0, 2, 3, 1).reshape(B, h * w, TOKEN[i])
    xn = _ln(prm, sd, p + "attn.norm1", 1e-5)                # the RC's own blocks use nn.LayerNorm defaults
    if TYPES[i] == "window":
        y = _window_attention(xn, h, w, sd, p + "attn.attn.", RC_HEADS[i], TOKEN[i])
    else:
        y = _token_attention(xn, sd, p + "attn.attn.", RC_HEADS[i], TOKEN[i])
    y = y + conv
    y = y + _mlp(_ln(y, sd, p + "attn.norm2", 1e-5), sd, p + "attn.mlp.")
    return y, (h, w)


def normal_cell(x, h, w, sd, i, b):
    """NormalCell.forward (NormalCell.py:155-236), class_token False, gamma False, shift 0."""
    p = PREFIX + "layers.%d.NC.%d." % (i, b)
    B, N, C = x.shape
    xn = _ln(x, sd, p + "norm1", 1e-6)                       # norm_layer = partial(nn.LayerNorm, eps=1e-6), vitae_v2.py:121
    if TYPES[i] == "window":
        a = _window_attention(xn, h, w, sd, p + "attn.", NC_HEADS[i], C)
    else:
        a = _full_attention(xn, sd, p + "attn.", NC_HEADS[i])
    conv = _pcm(x.view(B, h, w, C).permute(0, 3, 1, 2), sd, p + "PCM.", (1, 1, 1), NC_GROUP[i])
    x = x + a + conv.permute(0, 2, 3, 1).reshape(B, N, C)
    return x + _mlp(_ln(x, sd, p + "norm2", 1e-6), sd, p + "mlp.")


def vitae_v2_s(x, sd):
    """ViTAEv2.forward (vitae_v2.py:208-218): x [B,3,H,W] normalised -> {"stage3","stage4","stage5"} NCHW."""
    if x.shape[-2] % 32 or x.shape[-1] % 32:
        raise ValueError("ViTAEv2 needs inputs that are multiples of 32 (ReductionCell.py:143)")
    outs = {}
    B = x.shape[0]
    for i in range(4):
        t, (h, w) = reduction_cell(x, sd, i)
        for b in range(NC_DEPTH[i]):
            t = normal_cell(t, h, w, sd, i, b)
        x = t.view(B, h, w, -1).permute(0, 3, 1, 2)
        if i >= 1:
            outs["stage%d" % (i + 2)] = x.contiguous()
    return outs
