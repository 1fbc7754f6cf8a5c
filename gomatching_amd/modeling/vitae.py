"""ViTAEv2-S backbone driver (third_party/adet/modeling/vitae_v2/vitae_v2.py:98-249, built by the reference
through `build_vitaev2_backbone` with out_features stage3..5, channels 128 / 256 / 512).

Tokens stay channels-last [B*H*W, C].  Per stage one reduction cell (ReductionCell.py:133-188: dilated strided
convolutions -> concat -> attention, plus a parallel three-convolution branch on the cell's input) and NC_depth normal
cells (NormalCell.py:155-236: attention + parallel convolution branch + MLP).  Stages 1-2 use 7x7 window attention over a
centre-padded grid, stages 3-4 full attention over all tokens.  Every Linear, the dense convolutions and the two products
of the full attention are GEMM launches; the rest is csrc/vitae.hip.  Eval-mode semantics: DropPath / Dropout are
identities, BatchNorm2d is folded into the convolution epilogue, gradient checkpointing does not exist.

Input sizes must be multiples of 32 -- at any other size the reference's own ReductionCell fails (`assert N == H * W`,
ReductionCell.py:143 / the `.view(*x.shape)` at :165).
"""
import math

import torch

from .. import ops
from ..weights import VITAEV2_S

_f32 = torch.float32


class ViTAEv2S:
    out_features = ("stage3", "stage4", "stage5")
    strides = {"stage3": 8, "stage4": 16, "stage5": 32}
    channels = {"stage3": 128, "stage4": 256, "stage5": 512}
    size_divisibility = 32

    def __init__(self, sd, device, prefix="backbone.0.backbone."):
        self.device = device
        V = VITAEV2_S
        g = lambda k: sd[prefix + k].detach().float().contiguous().to(device)
        has = lambda k: (prefix + k) in sd
        lin = lambda k: (ops.prep_weight(g(k + ".weight")), g(k + ".bias") if has(k + ".bias") else None)
        ln = lambda k: (g(k + ".weight"), g(k + ".bias"))

        def pcm(p, groups, cin_pad=None):
            """conv-BN-SiLU, conv-BN-SiLU, conv: weights to OHWI, BatchNorm folded into (scale, shift) of the epilogue."""
            layers = []
            for conv, bn in (("0", "1"), ("3", "4"), ("6", None)):
                w = sd[prefix + p + conv + ".weight"].float().permute(0, 2, 3, 1)              # [Cout,3,3,Cin/groups]
                bias = sd[prefix + p + conv + ".bias"].float()
                if bn is None:
                    scale, shift = None, bias
                else:
                    q = lambda leaf: sd[prefix + p + bn + "." + leaf].float()
                    scale = q("weight") / torch.sqrt(q("running_var") + 1e-5)
                    shift = q("bias") + (bias - q("running_mean")) * scale
                if groups == 1:
                    if cin_pad and w.shape[-1] < cin_pad:                                       # the image is NHWC4
                        w = torch.cat([w, w.new_zeros(w.shape[:-1] + (cin_pad - w.shape[-1],))], -1)
                    wd = ops.prep_conv_weight(w.contiguous().to(device))
                else:
                    wd = w.permute(1, 2, 0, 3).contiguous().to(device)                          # [3,3,Cout,Cin/groups]
                layers.append((wd, None if scale is None else scale.contiguous().to(device), shift.contiguous().to(device)))
                cin_pad = None
            return layers

        self.stages = []
        cin = 4                                                  # the normalised image arrives as NHWC4 (4th channel zero)
        for i in range(4):
            E, T, k = V["embed"][i], V["token"][i], V["kernel"][i]
            p = "layers.%d.RC." % i
            prm = []
            for j, d in enumerate(V["dilations"][i]):
                w = sd[prefix + p + "PRM.convs.%d.0.weight" % j].float().permute(0, 2, 3, 1)   # [E,k,k,cin]
                if w.shape[-1] < cin:
                    w = torch.cat([w, w.new_zeros(w.shape[:-1] + (cin - w.shape[-1],))], -1)
                K = k * k * cin
                Kp = -(-K // 32) * 32
                wm = torch.zeros((E, Kp), dtype=_f32)
                wm[:, :K] = w.reshape(E, K)
                pad = math.ceil(((k - 1) * d + 1 - V["ratios"][i]) / 2)
                prm.append((ops.prep_weight(wm.to(device)), g(p + "PRM.convs.%d.0.bias" % j), d, pad, Kp))
            strides, res = [], V["ratios"][i] // 2
            for _ in range(3):
                strides.append((res > 0) + 1)
                res //= 2
            rc = {"prm": prm, "k": k, "ratio": V["ratios"][i], "pcm": pcm(p + "PCM.", V["rc_group"][i], cin_pad=cin),
                  "pcm_strides": strides, "group": V["rc_group"][i], "norm1": ln(p + "attn.norm1"),
                  "qkv": lin(p + "attn.attn.qkv"), "proj": lin(p + "attn.attn.proj"), "norm2": ln(p + "attn.norm2"),
                  "fc1": lin(p + "attn.mlp.fc1"), "fc2": lin(p + "attn.mlp.fc2"), "heads": V["rc_heads"][i]}
            ncs = []
            for b in range(V["nc_depth"][i]):
                p = "layers.%d.NC.%d." % (i, b)
                ncs.append({"norm1": ln(p + "norm1"), "qkv": lin(p + "attn.qkv"), "proj": lin(p + "attn.proj"),
                            "norm2": ln(p + "norm2"), "fc1": lin(p + "mlp.fc1"), "fc2": lin(p + "mlp.fc2"),
                            "pcm": pcm(p + "PCM.", V["nc_group"][i]), "group": V["nc_group"][i]})
            self.stages.append({"rc": rc, "nc": ncs, "E": E, "T": T, "type": V["tokens_type"][i],
                                "nc_heads": V["nc_heads"][i]})
            cin = T
        self._scratch = {}
        self.flash = True      # full attention through the fused kernel under the f16x3 back-end (else: GEMM pair + softmax)

    # ------------------------------------------------------------------------------------------------------------
    def _pcm(self, x, layers, groups, strides, R=None):
        """x [B,H,W,Cin] -> [B*OH*OW, Cout] (+R on the last convolution)."""
        for n, (w, scale, shift) in enumerate(layers):
            last = n == len(layers) - 1
            if groups == 1:
                Rv = None
                if last and R is not None:
                    Rv = R.view(x.shape[0], (x.shape[1] - 1) // strides[n] + 1, (x.shape[2] - 1) // strides[n] + 1, -1)
                x = ops.conv2d_nhwc(x, w, scale=scale, shift=shift, stride=strides[n], pad=1, R=Rv)
                if not last:
                    ops.silu_(x)
            else:
                x = ops.grouped_conv3x3(x, w, scale, shift, groups, stride=strides[n], silu=not last, R=R if last else None)
        return x.view(-1, x.shape[-1])

    def _zeros(self, key, shape):
        """Zero-initialised scratch whose padding columns are never written (so they stay zero)."""
        buf = self._scratch.get(key)
        if buf is None or tuple(buf.shape) != tuple(shape):
            buf = torch.zeros(shape, dtype=_f32, device=self.device)
            self._scratch[key] = buf
        return buf

    def _full_attention(self, qkv, B, N, heads):
        """softmax(q k^T / sqrt(hd)) v per (image, head).  Default (f16x3 back-end): one fused launch for all images and
        heads.  Otherwise two GEMM launches around a row softmax per (image, head); the score matrix is then
        [N, ceil4(N)] fp32 (205 MB at 64x112 tokens), reused across heads."""
        C = qkv.shape[1] // 3
        hd = C // heads
        if ops.GEMM_MODE == "f16x3" and self.flash:
            return ops.flash_attention(qkv, B, N, heads)         # scores stay on the CU (attn_flash.hip)
        Np = -(-N // 4) * 4
        S = self._zeros(("S", N), (N, Np))
        vt = self._zeros(("vt", N, hd), (hd, Np))
        out = torch.empty((B * N, C), dtype=_f32, device=self.device)
        split = ops.GEMM_MODE in ("f16x3", "bf16x6") and N >= 33
        for b in range(B):
            rows = qkv[b * N:(b + 1) * N]
            for h in range(heads):
                q = rows[:, h * hd:(h + 1) * hd]
                k = rows[:, C + h * hd:C + (h + 1) * hd]
                v = rows[:, 2 * C + h * hd:2 * C + (h + 1) * hd]
                # K and V^T are the "weight" operands of the two products: split at run time for the split back-ends
                ops.gemm(q, ops.split_weight(k) if split else k, out=S[:, :N])
                ops.softmax_rows_scaled_(S, N, hd ** -0.5)
                ops.transpose_into(v, vt)
                ops.gemm(S, ops.split_weight(vt) if split else vt, out=out[b * N:(b + 1) * N, h * hd:(h + 1) * hd])
        return out

    def _window_attention(self, xn, B, H, W, qkv, heads):
        win = ops.vitae_window_gather(xn, B, H, W)
        return ops.vitae_window_attention(ops.gemm(win, qkv[0], bias=qkv[1]), heads)

    def forward(self, x):
        """x: [B,H,W,4] normalised NHWC4.  Returns {"stage3","stage4","stage5"}: [B,h,w,C] channels-last."""
        B, H, W, _ = x.shape
        if H % 32 or W % 32:
            raise ValueError("the ViTAEv2 backbone needs inputs that are multiples of 32, got %dx%d (the reference's "
                             "ReductionCell asserts on anything else, ReductionCell.py:143)" % (H, W))
        outs = {}
        fmap = x.contiguous()
        for i, st in enumerate(self.stages):
            rc, T = st["rc"], st["T"]
            h, w = fmap.shape[1] // rc["ratio"], fmap.shape[2] // rc["ratio"]
            n_tok = h * w
            # -- reduction cell: PRM (dilated convolutions + GELU, concatenated by dilation)
            D = st["E"] * len(rc["prm"])
            prm = torch.empty((B * n_tok, D), dtype=_f32, device=self.device)
            for j, (wm, bias, d, pad, Kp) in enumerate(rc["prm"]):
                cols, oh, ow = ops.im2col(fmap, rc["k"], rc["k"], rc["ratio"], pad, d, Kp)
                assert (oh, ow) == (h, w)
                o = prm[:, j * st["E"]:(j + 1) * st["E"]]
                if isinstance(wm, ops.SplitWeight) and wm.kind == "f16x3":
                    ops.gemm(cols, wm, bias=bias, relu="gelu", out=o)
                else:
                    ops.gemm(cols, wm, bias=bias, out=o)
                del cols
            if not (isinstance(rc["prm"][0][0], ops.SplitWeight) and rc["prm"][0][0].kind == "f16x3"):
                ops.gelu_(prm)
            conv = self._pcm(fmap, rc["pcm"], rc["group"], rc["pcm_strides"])                   # [B*n_tok, T]
            xn = ops.layernorm_any(prm, *rc["norm1"], eps=1e-5)
            if st["type"] == "window":
                a = self._window_attention(xn, B, h, w, rc["qkv"], rc["heads"])
                a = ops.gemm(a, rc["proj"][0], bias=rc["proj"][1])
                t = ops.vitae_window_crop(a, B, h, w, R1=conv)
            else:
                qkv = ops.gemm(xn, rc["qkv"][0], bias=rc["qkv"][1])
                a = self._full_attention(qkv, B, n_tok, rc["heads"])
                # token_transformer.py:39-42: the skip connection is V (input and output widths differ)
                v = qkv[:, 2 * T:]
                t = ops.gemm(a, rc["proj"][0], bias=rc["proj"][1], R=v)
                t = ops.add(t, conv)
            hdn = ops.gemm_gelu(ops.layernorm_any(t, *rc["norm2"], eps=1e-5), rc["fc1"][0], bias=rc["fc1"][1])
            t = ops.gemm(hdn, rc["fc2"][0], bias=rc["fc2"][1], R=t)
            # -- normal cells
            for nc in st["nc"]:
                conv = self._pcm(t.view(B, h, w, T), nc["pcm"], nc["group"], (1, 1, 1), R=t)    # shortcut + convX
                xn = ops.layernorm_any(t, *nc["norm1"], eps=1e-6)
                if st["type"] == "window":
                    a = self._window_attention(xn, B, h, w, nc["qkv"], st["nc_heads"])
                    a = ops.gemm(a, nc["proj"][0], bias=nc["proj"][1])
                    t = ops.vitae_window_crop(a, B, h, w, R1=conv)
                else:
                    qkv = ops.gemm(xn, nc["qkv"][0], bias=nc["qkv"][1])
                    a = self._full_attention(qkv, B, n_tok, st["nc_heads"])
                    t = ops.gemm(a, nc["proj"][0], bias=nc["proj"][1], R=conv)
                hdn = ops.gemm_gelu(ops.layernorm_any(t, *nc["norm2"], eps=1e-6), nc["fc1"][0], bias=nc["fc1"][1])
                t = ops.gemm(hdn, nc["fc2"][0], bias=nc["fc2"][1], R=t)
            fmap = t.view(B, h, w, T)
            if i >= 1:
                outs["stage%d" % (i + 2)] = fmap
        return outs
