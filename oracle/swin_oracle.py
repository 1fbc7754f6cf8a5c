"""CPU restatement of the Swin-T backbone of the reference (third_party/adet/modeling/swin/swin_transformer.py).
TEST INFRASTRUCTURE: imported only by tests/ and the fixture generator.

Functional torch-CPU code over the reference's own state-dict key names (prefix `backbone.0.backbone.`), eval mode
(DropPath / Dropout are identities).  PINNED by tests/golden/swin_tiny.npz, which oracle/gen_golden_swin.py produces by
running the reference's unmodified `SwinTransformer` module on the repo's synthetic weights.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

DEPTHS, HEADS, EMBED, WINDOW = (2, 2, 6, 2), (3, 6, 12, 24), 96, 7
PREFIX = "backbone.0.backbone."


def relative_position_index(ws=WINDOW):
    """swin_transformer.py:111-123."""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def shift_mask(H, W, ws=WINDOW):
    """BasicLayer.forward mask (:411-434): [nW, ws*ws, ws*ws] of 0 / -100 for the padded Hp x Wp grid."""
    Hp, Wp = int(np.ceil(H / ws)) * ws, int(np.ceil(W / ws)) * ws
    img = torch.zeros((1, Hp, Wp, 1))
    sl = (slice(0, -ws), slice(-ws, -(ws // 2)), slice(-(ws // 2), None))
    cnt = 0
    for h in sl:
        for w in sl:
            img[:, h, w, :] = cnt
            cnt += 1
    mw = img.view(1, Hp // ws, ws, Wp // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def _block(x, H, W, sd, p, heads, shift, mask):
    """SwinTransformerBlock.forward (:233-291) + WindowAttention.forward (:133-167)."""
    B, L, C = x.shape
    ws = WINDOW
    shortcut = x
    x = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"]).view(B, H, W, C)
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
    Hp, Wp = H + pad_b, W + pad_r
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    B_, N = xw.shape[0], ws * ws
    qkv = F.linear(xw, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B_, N, 3, heads, C // heads)
    qkv = qkv.permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // heads) ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = sd[p + "attn.relative_position_bias_table"][relative_position_index().view(-1)].view(N, N, -1)
    attn = attn + bias.permute(2, 0, 1).unsqueeze(0)
    if shift > 0:
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    attn = attn.softmax(-1)
    xw = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    xw = F.linear(xw, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    x = xw.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    x = x[:, :H, :W, :].reshape(B, H * W, C)
    x = shortcut + x
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def _merge(x, H, W, sd, p):
    """PatchMerging.forward (:308-332)."""
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    if H % 2 or W % 2:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    x = x.view(B, -1, 4 * C)
    x = F.layer_norm(x, (4 * C,), sd[p + "norm.weight"], sd[p + "norm.bias"])
    return F.linear(x, sd[p + "reduction.weight"])


def swin_tiny(x, sd, prefix=PREFIX, depths=None):
    """x: [B,3,H,W] normalised.  Returns {'stage3','stage4','stage5'} NCHW (strides 8, 16, 32), SwinTransformer.forward
    (:682-...) with out_features stage3..5 and patch_norm."""
    g = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    _, _, H, W = x.shape
    if W % 4:
        x = F.pad(x, (0, 4 - W % 4))
    if H % 4:
        x = F.pad(x, (0, 0, 0, 4 - H % 4))
    x = F.conv2d(x, g["patch_embed.proj.weight"], g["patch_embed.proj.bias"], stride=4)
    Wh, Ww = x.shape[2], x.shape[3]
    x = x.flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (EMBED,), g["patch_embed.norm.weight"], g["patch_embed.norm.bias"])
    outs = {}
    if depths is None:                                           # Swin-S = the same network with 18 stage-3 blocks (:709-721)
        n3 = sum(1 for k in g if k.startswith("layers.2.blocks.") and k.endswith(".norm1.weight"))
        depths = DEPTHS[:2] + (n3,) + DEPTHS[3:]
    for i, (depth, heads) in enumerate(zip(depths, HEADS)):
        C = EMBED * 2 ** i
        mask = shift_mask(Wh, Ww)
        for b in range(depth):
            x = _block(x, Wh, Ww, g, "layers.%d.blocks.%d." % (i, b), heads, 0 if b % 2 == 0 else WINDOW // 2, mask)
        if i >= 1:
            o = F.layer_norm(x, (C,), g["norm%d.weight" % i], g["norm%d.bias" % i])
            outs["stage%d" % (i + 2)] = o.view(-1, Wh, Ww, C).permute(0, 3, 1, 2).contiguous()
        if i < len(DEPTHS) - 1:
            x = _merge(x, Wh, Ww, g, "layers.%d.downsample." % i)
            Wh, Ww = (Wh + 1) // 2, (Ww + 1) // 2
    return outs
