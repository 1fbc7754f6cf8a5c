"""Swin-T / Swin-S backbone driver (third_party/adet/modeling/swin/swin_transformer.py:491-724, built by the
reference through `build_swin_backbone` with out_features stage3..5).

Tokens stay channels-last [B*H*W, C]; every Linear is one GEMM launch (bias / residual / in the epilogue), the window
logic is three gather kernels and one attention kernel (csrc/swin.hip).  Eval-mode semantics: DropPath and Dropout
are identities.  The relative-position bias [heads,49,49] of every block is gathered from its table once at load;
the shifted-window mask is built once per map size.
"""
import numpy as np
import torch

from .. import ops
from ..weights import SWIN_TYPES

_f32 = torch.float32


def _relative_position_index(ws):
    coords = np.stack(np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij")).reshape(2, -1)
    rel = (coords[:, :, None] - coords[:, None, :]).transpose(1, 2, 0).copy()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _shift_mask(H, W, ws):
    """BasicLayer.forward :411-434 for the padded grid: [nW, 49, 49] of 0 / -100."""
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    img = np.zeros((Hp, Wp), np.float32)
    sl = (slice(0, -ws), slice(-ws, -(ws // 2)), slice(-(ws // 2), None))
    cnt = 0
    for h in sl:
        for w in sl:
            img[h, w] = cnt
            cnt += 1
    mw = img.reshape(Hp // ws, ws, Wp // ws, ws).transpose(0, 2, 1, 3).reshape(-1, ws * ws)
    d = mw[:, None, :] - mw[:, :, None]
    return np.where(d != 0, np.float32(-100.0), np.float32(0.0))


class SwinTiny:
    out_features = ("stage3", "stage4", "stage5")
    strides = {"stage3": 8, "stage4": 16, "stage5": 32}
    channels = {"stage3": 192, "stage4": 384, "stage5": 768}
    size_divisibility = 32          # what the reference's module reports (:651); GoMatching never pads to it

    def __init__(self, sd, device, prefix="backbone.0.backbone.", swin_type="tiny"):
        self.device = device
        S = SWIN_TYPES[swin_type]                                # tiny | small: the same blocks, stage-3 depth 6 | 18
        self.ws = S["window"]
        g = lambda k: sd[prefix + k].detach().float().contiguous().to(device)
        lin = lambda k: (ops.prep_weight(g(k + ".weight")), g(k + ".bias") if (prefix + k + ".bias") in sd else None)
        w = sd[prefix + "patch_embed.proj.weight"].float()                  # [96,3,4,4] -> [96, kh, kw, c(4)]
        w = torch.cat([w.permute(0, 2, 3, 1), w.new_zeros(w.shape[0], 4, 4, 1)], -1).reshape(w.shape[0], 64)
        self.patch = (ops.prep_weight(w.contiguous().to(device)), g("patch_embed.proj.bias"),
                      g("patch_embed.norm.weight"), g("patch_embed.norm.bias"))
        idx = torch.from_numpy(_relative_position_index(self.ws).reshape(-1))
        self.stages = []
        for i, (depth, heads) in enumerate(zip(S["depths"], S["heads"])):
            blocks = []
            for b in range(depth):
                p = "layers.%d.blocks.%d." % (i, b)
                table = sd[prefix + p + "attn.relative_position_bias_table"].float()
                bias = table[idx].view(49, 49, heads).permute(2, 0, 1).contiguous().to(device)
                blocks.append({"norm1": (g(p + "norm1.weight"), g(p + "norm1.bias")), "qkv": lin(p + "attn.qkv"),
                               "proj": lin(p + "attn.proj"), "bias": bias,
                               "norm2": (g(p + "norm2.weight"), g(p + "norm2.bias")), "fc1": lin(p + "mlp.fc1"),
                               "fc2": lin(p + "mlp.fc2"), "shift": 0 if b % 2 == 0 else self.ws // 2})
            st = {"blocks": blocks, "heads": heads, "C": S["embed"] * 2 ** i}
            if i < len(S["depths"]) - 1:
                p = "layers.%d.downsample." % i
                st["merge"] = (g(p + "norm.weight"), g(p + "norm.bias"), ops.prep_weight(g(p + "reduction.weight")))
            if i >= 1:
                st["norm"] = (g("norm%d.weight" % i), g("norm%d.bias" % i))
            self.stages.append(st)
        self._masks = {}

    def _mask(self, H, W):
        key = (H, W)
        if key not in self._masks:
            self._masks[key] = torch.from_numpy(_shift_mask(H, W, self.ws)).contiguous().to(self.device)
        return self._masks[key]

    def forward(self, x):
        """x: [B,H,W,4] normalised NHWC4 (as `ops.preprocess` / `ops.ingest` produce).  Returns
        {"stage3","stage4","stage5"}: [B,h,w,C] channels-last, strides 8 / 16 / 32."""
        B, H, W, _ = x.shape
        rows, Hc, Wc = ops.swin_patchify(x.contiguous())
        w, b, gn, bn = self.patch
        t = ops.layernorm_any(ops.gemm(rows, w, bias=b), gn, bn)
        outs = {}
        for i, st in enumerate(self.stages):
            C, heads = st["C"], st["heads"]
            nW = (-(-Hc // self.ws)) * (-(-Wc // self.ws))
            for blk in st["blocks"]:
                xn = ops.layernorm_any(t, *blk["norm1"])
                win = ops.swin_window_gather(xn, B, Hc, Wc, blk["shift"])
                qkv = ops.gemm(win, blk["qkv"][0], bias=blk["qkv"][1])
                att = ops.swin_window_attention(qkv, blk["bias"], self._mask(Hc, Wc) if blk["shift"] else None, nW, heads)
                att = ops.gemm(att, blk["proj"][0], bias=blk["proj"][1])
                t = ops.swin_window_scatter_add(att, t, B, Hc, Wc, blk["shift"])
                h = ops.gemm_gelu(ops.layernorm_any(t, *blk["norm2"]), blk["fc1"][0], bias=blk["fc1"][1])
                t = ops.gemm(h, blk["fc2"][0], bias=blk["fc2"][1], R=t)
            if "norm" in st:
                outs["stage%d" % (i + 2)] = ops.layernorm_any(t, *st["norm"]).view(B, Hc, Wc, C)
            if "merge" in st:
                gm, bm, wm = st["merge"]
                m, Hc, Wc = ops.swin_patch_merge(t, B, Hc, Wc)
                t = ops.gemm(ops.layernorm_any(m, gm, bm), wm)
        return outs
