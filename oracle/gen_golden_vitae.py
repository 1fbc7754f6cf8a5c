"""Generate tests/golden/vitae_s.npz by running the REFERENCE's own ViTAEv2 module (container-only).

TEST INFRASTRUCTURE.  Run in the build container, where /root/reference exists:

    python -m oracle.gen_golden_vitae

third_party/adet/modeling/vitae_v2/*.py are imported unmodified behind the same sys.modules stand-ins as the Swin
generator (timm's DropPath / to_2tuple / trunc_normal_, Detectron2's Backbone / registry / ShapeSpec), built with the
constructor arguments of `build_vitaev2_backbone` (vitae_v2.py:228-249), loaded with the repo's synthetic weights and
executed on CPU (use_checkpoint is a training-memory device and is switched off); only inputs and outputs are committed.
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from gomatching_amd.config import setup_cfg                      # noqa: E402
from gomatching_amd.weights import synth_state_dict              # noqa: E402
from oracle import gen_golden_swin, ref_shim, vitae_oracle       # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def build_reference(sd):
    gen_golden_swin._stubs()
    import numpy.core.fromnumeric                               # noqa: F401  (ReductionCell.py:2 imports from it)
    ref_shim._pkg("adet.modeling.vitae_v2", ref_shim.REF_ROOT + "/third_party/adet/modeling/vitae_v2")
    mod = importlib.import_module("adet.modeling.vitae_v2.vitae_v2")
    net = mod.ViTAEv2(in_chans=3, RC_tokens_type=['window', 'window', 'transformer', 'transformer'],
                      NC_tokens_type=['window', 'window', 'transformer', 'transformer'], embed_dims=[64, 64, 128, 256],
                      token_dims=[64, 128, 256, 512], downsample_ratios=[4, 2, 2, 2], NC_depth=[2, 2, 8, 2],
                      NC_heads=[1, 2, 4, 8], RC_heads=[1, 1, 2, 4], mlp_ratio=4., NC_group=[1, 32, 64, 128],
                      RC_group=[1, 16, 32, 64], use_checkpoint=False, drop_rate=0., attn_drop_rate=0., window_size=7,
                      drop_path_rate=0.2)
    prefix = "backbone.0.backbone."
    own = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    ref_keys = {k: tuple(v.shape) for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}
    mine = {k: tuple(v.shape) for k, v in own.items()}
    assert ref_keys == mine, (sorted(set(ref_keys) ^ set(mine))[:10],
                              [(k, ref_keys[k], mine[k]) for k in ref_keys if k in mine and ref_keys[k] != mine[k]][:10])
    missing, unexpected = net.load_state_dict(own, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    return net.eval()


def main():
    cfg = setup_cfg(builtin="icdar15")
    cfg.MODEL.BACKBONE.NAME = "build_vitaev2_backbone"
    sd = synth_state_dict(cfg, seed=3)
    net = build_reference(sd)
    out = {}
    g = torch.Generator().manual_seed(1)
    for tag, (B, H, W) in {"a": (2, 96, 128), "b": (1, 160, 96), "c": (1, 64, 224)}.items():
        x = torch.randn(B, 3, H, W, generator=g)
        with torch.no_grad():
            ref = net(x)
            mine = vitae_oracle.vitae_v2_s(x, sd)
        out["x_" + tag] = x.numpy()
        for k in ("stage3", "stage4", "stage5"):
            out["%s_%s" % (k, tag)] = ref[k].numpy()
            print(tag, k, tuple(ref[k].shape), "scale %.2f" % float(ref[k].abs().max()),
                  "oracle vs reference max|d| = %.2e" % float((ref[k] - mine[k]).abs().max()))
    np.savez_compressed(os.path.join(GOLD, "vitae_s.npz"), **out)
    print("wrote", os.path.join(GOLD, "vitae_s.npz"), os.path.getsize(os.path.join(GOLD, "vitae_s.npz")))


if __name__ == "__main__":
    main()
