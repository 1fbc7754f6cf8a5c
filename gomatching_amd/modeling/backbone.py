"""ResNet-50 / FrozenBN backbone driver: every convolution is one launch of the
implicit-GEMM MFMA kernel with FrozenBatchNorm, ReLU and the bottleneck shortcut fused into its epilogue.

Follows Detectron2 v0.6 `build_resnet_backbone` for DEPTH 50, STRIDE_IN_1X1 False, FrozenBN (the backbone
the reference builds at gom_lstmatcher.py:46 from configs/*.yaml:6-11): stem conv7x7/2 + BN + ReLU +
maxpool3x3/2, bottlenecks [3,4,6,3] with the stride on the 3x3, 1x1 projection shortcuts, outputs
res3/res4/res5.  Activations are channels-last so a 1x1 convolution is a plain GEMM over pixels.
"""
import torch

from .. import ops

_STAGES = (("res2", 3, 1), ("res3", 4, 2), ("res4", 6, 2), ("res5", 3, 2))


def _fold_bn(sd, name, device, eps=1e-5):
    # same arithmetic as FrozenBatchNorm2d.forward: scale = w * rsqrt(var + eps); shift = b - mean * scale
    scale = sd[name + ".norm.weight"] * (sd[name + ".norm.running_var"] + eps).rsqrt()
    shift = sd[name + ".norm.bias"] - sd[name + ".norm.running_mean"] * scale
    return scale.float().contiguous().to(device), shift.float().contiguous().to(device)


def _ohwi(w, device, pad_cin_to=None):
    w = w.float().permute(0, 2, 3, 1)
    if pad_cin_to is not None and w.shape[-1] < pad_cin_to:
        w = torch.cat([w, w.new_zeros(w.shape[:-1] + (pad_cin_to - w.shape[-1],))], -1)
    return w.contiguous().to(device)


class ResNet50:
    out_features = ("res3", "res4", "res5")
    strides = {"res3": 8, "res4": 16, "res5": 32}
    channels = {"res3": 512, "res4": 1024, "res5": 2048}

    def __init__(self, sd, device, prefix="backbone.0.backbone."):
        self.device = device
        self.convs = {}

        def add(name, pad_cin_to=None):
            w = ops.prep_conv_weight(_ohwi(sd[prefix + name + ".weight"], device, pad_cin_to))
            sc, sh = _fold_bn(sd, prefix + name, device)
            self.convs[name] = (w, sc, sh)

        add("stem.conv1", pad_cin_to=4)
        self.blocks = []
        for stage, nblk, first_stride in _STAGES:
            for i in range(nblk):
                p = "%s.%d." % (stage, i)
                has_sc = (prefix + p + "shortcut.weight") in sd
                if has_sc:
                    add(p + "shortcut")
                for c in ("conv1", "conv2", "conv3"):
                    add(p + c)
                self.blocks.append((stage, p, first_stride if i == 0 else 1, has_sc, i == nblk - 1))
        # conv3 + residual + ReLU of block k fused with conv1 of block k + 1 (csrc/bneck_fused.hip) where the shapes are served
        # (res2, res3, inside res4) under the f16x3 back-end: block k's output is written once and never read back by conv1
        self.fused = {}
        for (_, p, _, _, _), (_, pn, _, _, _) in zip(self.blocks[:-1], self.blocks[1:]):
            w3, sc3, sh3 = self.convs[p + "conv3"]
            w1, sc1, sh1 = self.convs[pn + "conv1"]
            if ops.BneckFused.serves(w3, w1):
                self.fused[p] = ops.BneckFused(w3, sc3, sh3, w1, sc1, sh1)

    def _conv(self, x, name, stride=1, pad=0, relu=False, R=None):
        w, sc, sh = self.convs[name]
        return ops.conv2d_nhwc(x, w, scale=sc, shift=sh, R=R, relu=relu, stride=stride, pad=pad)

    def forward(self, x_nhwc4):
        """x: [B,H,W,4] normalised RGB + zero channel.  Returns {'res3','res4','res5'} NHWC."""
        w, sc, sh = self.convs["stem.conv1"]
        if ops.stem_pool_serves(w):                          # one launch: the 910 MB between conv and pool never exist
            x = ops.stem_conv_pool(x_nhwc4.contiguous(), w, scale=sc, shift=sh)
        else:
            x = self._conv(x_nhwc4, "stem.conv1", stride=2, pad=3, relu=True)
            x = ops.maxpool3x3s2(x)
        outs = {}
        y1 = None                                            # conv1 of this block, when the previous block's launch made it
        for stage, p, s, has_sc, last in self.blocks:
            sc = self._conv(x, p + "shortcut", stride=s) if has_sc else x
            y = y1 if y1 is not None else self._conv(x, p + "conv1", relu=True)
            y = self._conv(y, p + "conv2", stride=s, pad=1, relu=True)
            blk = self.fused.get(p)
            if blk is not None:
                x, y1 = ops.bneck_fused(y, blk, sc)                   # relu(conv3 + shortcut) and the next block's conv1
            else:
                x, y1 = self._conv(y, p + "conv3", relu=True, R=sc), None
            if last:
                outs[stage] = x
        return {k: outs[k] for k in self.out_features}
