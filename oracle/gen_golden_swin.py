"""Generate tests/golden/swin_tiny.npz by running the REFERENCE's own SwinTransformer module (container-only).

TEST INFRASTRUCTURE.  Run in the build container, where /root/reference exists:

    python -m oracle.gen_golden_swin

third_party/adet/modeling/swin/swin_transformer.py is imported unmodified behind sys.modules stand-ins for the packages
this image lacks (timm's DropPath / to_2tuple / trunc_normal_, Detectron2's Backbone base class and registry), loaded with
the repo's synthetic weights and executed on CPU; only inputs and outputs are committed.
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from gomatching_amd.config import setup_cfg                      # noqa: E402
from gomatching_amd.weights import synth_state_dict              # noqa: E402
from oracle import ref_shim, swin_oracle                         # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _stubs():
    ref_shim.install()

    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class DropPath(nn.Module):                                    # eval mode: identity
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    mod("timm")
    mod("timm.models")
    mod("timm.models.layers", DropPath=DropPath, to_2tuple=lambda x: x if isinstance(x, tuple) else (x, x),
        trunc_normal_=lambda t, std=1.0: nn.init.trunc_normal_(t, std=std))

    class Backbone(nn.Module):
        pass

    class _Reg:
        def register(self, obj=None):
            return obj if obj is not None else (lambda o: o)

    mod("detectron2.modeling.backbone", Backbone=Backbone)
    mod("detectron2.modeling.backbone.build", BACKBONE_REGISTRY=_Reg())
    _pkg = ref_shim._pkg
    _pkg("adet.modeling.swin", ref_shim.REF_ROOT + "/third_party/adet/modeling/swin")


def main():
    _stubs()
    import importlib
    swin_mod = importlib.import_module("adet.modeling.swin.swin_transformer")
    cfg = setup_cfg(builtin="icdar15")
    cfg.MODEL.BACKBONE.NAME = "build_swin_backbone"
    sd = synth_state_dict(cfg, seed=3)
    prefix = "backbone.0.backbone."
    net = swin_mod.SwinTransformer(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=7,
                                   mlp_ratio=4, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                                   drop_path_rate=0.2, ape=False, patch_norm=True, frozen_stages=-1,
                                   out_features=["stage3", "stage4", "stage5"])
    own = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    missing, unexpected = net.load_state_dict(own, strict=False)
    assert not unexpected and all("relative_position_index" in k for k in missing), (missing, unexpected)
    net.eval()
    out = {}
    g = torch.Generator().manual_seed(1)
    for tag, (B, H, W) in {"a": (2, 96, 128), "b": (1, 90, 130), "c": (1, 224, 64)}.items():
        x = torch.randn(B, 3, H, W, generator=g)
        with torch.no_grad():
            ref = net(x)
            mine = swin_oracle.swin_tiny(x, sd)
        out["x_" + tag] = x.numpy()
        for k in ("stage3", "stage4", "stage5"):
            out["%s_%s" % (k, tag)] = ref[k].numpy()
            print(tag, k, tuple(ref[k].shape), "oracle vs reference max|d| = %.2e" % float((ref[k] - mine[k]).abs().max()))
    if "--small-only" not in sys.argv:
        np.savez_compressed(os.path.join(GOLD, "swin_tiny.npz"), **out)
        print("wrote", os.path.join(GOLD, "swin_tiny.npz"), os.path.getsize(os.path.join(GOLD, "swin_tiny.npz")))
    # Swin-S (swin_transformer.py:709-721: depths (2, 2, 18, 2), everything else as tiny), one odd-sized input
    cfg.MODEL.SWIN.TYPE = "small"
    sd = synth_state_dict(cfg, seed=4)
    net = swin_mod.SwinTransformer(embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24), window_size=7,
                                   mlp_ratio=4, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                                   drop_path_rate=0.3, ape=False, patch_norm=True, frozen_stages=-1,
                                   out_features=["stage3", "stage4", "stage5"])
    own = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    missing, unexpected = net.load_state_dict(own, strict=False)
    assert not unexpected and all("relative_position_index" in k for k in missing), (missing, unexpected)
    net.eval()
    x = torch.randn(1, 3, 90, 130, generator=g)
    with torch.no_grad():
        ref = net(x)
        mine = swin_oracle.swin_tiny(x, sd)
    out = {"x_s": x.numpy()}
    for k in ("stage3", "stage4", "stage5"):
        out["%s_s" % k] = ref[k].numpy()
        print("small", k, tuple(ref[k].shape), "oracle vs reference max|d| = %.2e" % float((ref[k] - mine[k]).abs().max()))
    np.savez_compressed(os.path.join(GOLD, "swin_small.npz"), **out)
    print("wrote", os.path.join(GOLD, "swin_small.npz"), os.path.getsize(os.path.join(GOLD, "swin_small.npz")))


if __name__ == "__main__":
    main()
