"""State-dict handling for the GoMatching inference path.

* `canonical_keys(cfg)` — every parameter/buffer name (reference naming, SURVEY.md §8-b "Weights")
  with its shape, for R-50/FrozenBN + DeepSolo-without-backbone + the matcher head.
* `synth_state_dict(cfg, seed)` — deterministic synthetic weights from a counter-based generator
  (numpy Philox keyed by crc32(name) ^ seed), so the GPU box regenerates bit-identical weights
  without shipping a checkpoint ("random-init weights of that architecture", bench contract).
* `normalize_state_dict(sd)` — accepts reference checkpoints: applies the rename rule of
  /root/reference/tools/decouple_deepsolo.py:13-19 (`detection_transformer.backbone.*` ->
  `backbone.*`), strips a leading `model.` wrapper, and resolves the shared-head aliases
  (/root/reference/third_party/adet/modeling/model/detection_transformer_wobackbone.py:128-155).
"""
import math
import zlib

import numpy as np

_R50_STAGES = (("res2", 3, 64, 256), ("res3", 4, 128, 512), ("res4", 6, 256, 1024), ("res5", 3, 512, 2048))


def _resnet_keys(prefix="backbone.0.backbone."):
    keys = {}

    def conv(name, cout, cin, k):
        keys[prefix + name + ".weight"] = (cout, cin, k, k)
        for s in ("weight", "bias", "running_mean", "running_var"):
            keys[prefix + name + ".norm." + s] = (cout,)

    conv("stem.conv1", 64, 3, 7)
    cin = 64
    for stage, nblk, mid, cout in _R50_STAGES:
        for i in range(nblk):
            p = "%s.%d." % (stage, i)
            if i == 0:
                conv(p + "shortcut", cout, cin, 1)
            conv(p + "conv1", mid, cin, 1)
            conv(p + "conv2", mid, mid, 3)
            conv(p + "conv3", cout, mid, 1)
            cin = cout
    return keys


SWIN_TINY = {"embed": 96, "depths": (2, 2, 6, 2), "heads": (3, 6, 12, 24), "window": 7}
# the two types whose channels DeepSolo's input_proj admits (detection_transformer_wobackbone.py:59-62); swin_transformer.py:696-721
SWIN_TYPES = {"tiny": SWIN_TINY, "small": dict(SWIN_TINY, depths=(2, 2, 18, 2))}


def _swin_keys(prefix="backbone.0.backbone.", swin_type="tiny"):
    """Swin-T / Swin-S, out_features stage3..5, patch_norm (swin_transformer.py:692-724)."""
    S = SWIN_TYPES[swin_type]
    keys = {prefix + "patch_embed.proj.weight": (S["embed"], 3, 4, 4), prefix + "patch_embed.proj.bias": (S["embed"],)}
    _ln(keys, prefix + "patch_embed.norm", S["embed"])
    for i, (depth, heads) in enumerate(zip(S["depths"], S["heads"])):
        C = S["embed"] * 2 ** i
        for b in range(depth):
            p = prefix + "layers.%d.blocks.%d." % (i, b)
            _ln(keys, p + "norm1", C)
            keys[p + "attn.relative_position_bias_table"] = ((2 * S["window"] - 1) ** 2, heads)
            _lin(keys, p + "attn.qkv", 3 * C, C)
            _lin(keys, p + "attn.proj", C, C)
            _ln(keys, p + "norm2", C)
            _lin(keys, p + "mlp.fc1", 4 * C, C)
            _lin(keys, p + "mlp.fc2", C, 4 * C)
        if i < len(S["depths"]) - 1:
            p = prefix + "layers.%d.downsample." % i
            keys[p + "reduction.weight"] = (2 * C, 4 * C)
            _ln(keys, p + "norm", 4 * C)
        if i >= 1:
            _ln(keys, prefix + "norm%d" % i, C)
    return keys


VITAEV2_S = {"embed": (64, 64, 128, 256), "token": (64, 128, 256, 512), "ratios": (4, 2, 2, 2), "kernel": (7, 3, 3, 3),
             "dilations": ((1, 2, 3, 4), (1, 2, 3), (1, 2), (1, 2)), "rc_heads": (1, 1, 2, 4), "nc_heads": (1, 2, 4, 8),
             "rc_group": (1, 16, 32, 64), "nc_group": (1, 32, 64, 128), "nc_depth": (2, 2, 8, 2),
             "tokens_type": ("window", "window", "transformer", "transformer"), "window": 7, "mlp_ratio": 4}


def _bn(keys, name, c):
    for leaf in ("weight", "bias", "running_mean", "running_var"):
        keys[name + "." + leaf] = (c,)


def _gconv(keys, name, cout, cin, groups, k=3):
    keys[name + ".weight"] = (cout, cin // groups, k, k)
    keys[name + ".bias"] = (cout,)


def _vitae_keys(prefix="backbone.0.backbone."):
    """ViTAEv2-S as `build_vitaev2_backbone` builds it (vitae_v2.py:228-249): per stage one ReductionCell
    (ReductionCell.py:66-131) and NC_depth NormalCells (NormalCell.py:113-153)."""
    V = VITAEV2_S
    keys = {}
    cin = 3
    for i in range(4):
        E, T = V["embed"][i], V["token"][i]
        p = prefix + "layers.%d.RC." % i
        for j in range(len(V["dilations"][i])):
            keys[p + "PRM.convs.%d.0.weight" % j] = (E, cin, V["kernel"][i], V["kernel"][i])
            keys[p + "PRM.convs.%d.0.bias" % j] = (E,)
        g = V["rc_group"][i]
        _gconv(keys, p + "PCM.0", E, cin, g); _bn(keys, p + "PCM.1", E)
        _gconv(keys, p + "PCM.3", E, E, g); _bn(keys, p + "PCM.4", E)
        _gconv(keys, p + "PCM.6", T, E, g)
        D = E * len(V["dilations"][i])                      # op='cat'
        _ln(keys, p + "attn.norm1", D)
        _lin(keys, p + "attn.attn.qkv", 3 * T, D)
        if V["tokens_type"][i] == "transformer":            # Token_transformer's Attention: qkv_bias=False (:12)
            del keys[p + "attn.attn.qkv.bias"]
        _lin(keys, p + "attn.attn.proj", T, T)
        _ln(keys, p + "attn.norm2", T)
        _lin(keys, p + "attn.mlp.fc1", T, T)                # ReductionCell's mlp_ratio is its default 1.0
        _lin(keys, p + "attn.mlp.fc2", T, T)
        g = V["nc_group"][i]
        Hd = T * V["mlp_ratio"]
        for b in range(V["nc_depth"][i]):
            p = prefix + "layers.%d.NC.%d." % (i, b)
            _ln(keys, p + "norm1", T)
            _lin(keys, p + "attn.qkv", 3 * T, T)
            _lin(keys, p + "attn.proj", T, T)
            _ln(keys, p + "norm2", T)
            _lin(keys, p + "mlp.fc1", Hd, T)
            _lin(keys, p + "mlp.fc2", T, Hd)
            _gconv(keys, p + "PCM.0", Hd, T, g); _bn(keys, p + "PCM.1", Hd)
            _gconv(keys, p + "PCM.3", T, Hd, g); _bn(keys, p + "PCM.4", T)
            _gconv(keys, p + "PCM.6", T, T, g)
        cin = T
    return keys


def backbone_channels(cfg):
    """Channel table of detection_transformer_wobackbone.py:59-70."""
    name = cfg.MODEL.BACKBONE.NAME
    if name == "build_swin_backbone":
        return [192, 384, 768]
    if name == "build_resnet_backbone":
        return [512, 1024, 2048]
    if name == "build_vitaev2_backbone":
        if cfg.MODEL.ViTAEv2.TYPE != "vitaev2_s":
            raise NotImplementedError("only vitaev2_s exists (detection_transformer_wobackbone.py:64-68)")
        return [128, 256, 512]
    raise NotImplementedError("backbone %s is not built (SURVEY.md §8-f3)" % name)


def _lin(keys, name, cout, cin):
    keys[name + ".weight"] = (cout, cin)
    keys[name + ".bias"] = (cout,)


def _ln(keys, name, d):
    keys[name + ".weight"] = (d,)
    keys[name + ".bias"] = (d,)


def _mlp(keys, name, din, dh, dout, n):
    dims = [din] + [dh] * (n - 1) + [dout]
    for i in range(n):
        _lin(keys, "%s.layers.%d" % (name, i), dims[i + 1], dims[i])


def _mha(keys, name, d):
    keys[name + ".in_proj_weight"] = (3 * d, d)
    keys[name + ".in_proj_bias"] = (3 * d,)
    _lin(keys, name + ".out_proj", d, d)


def _msda(keys, name, d, heads, levels, points):
    _lin(keys, name + ".sampling_offsets", heads * levels * points * 2, d)
    _lin(keys, name + ".attention_weights", heads * levels * points, d)
    _lin(keys, name + ".value_proj", d, d)
    _lin(keys, name + ".output_proj", d, d)


def _deepsolo_keys(cfg, prefix="detection_transformer."):
    T = cfg.MODEL.TRANSFORMER
    d, heads, L = T.HIDDEN_DIM, T.NHEADS, T.NUM_FEATURE_LEVELS
    ffn = T.DIM_FEEDFORWARD
    keys = {}
    chans = backbone_channels(cfg)
    for l in range(3):
        keys[prefix + "input_proj.%d.0.weight" % l] = (d, chans[l], 1, 1)
        keys[prefix + "input_proj.%d.0.bias" % l] = (d,)
        _ln(keys, prefix + "input_proj.%d.1" % l, d)
    keys[prefix + "input_proj.3.0.weight"] = (d, chans[-1], 3, 3)
    keys[prefix + "input_proj.3.0.bias"] = (d,)
    _ln(keys, prefix + "input_proj.3.1", d)
    keys[prefix + "point_embed.weight"] = (T.NUM_QUERIES * T.NUM_POINTS, d)
    t = prefix + "transformer."
    keys[t + "level_embed"] = (L, d)
    _lin(keys, t + "enc_output", d, d)
    _ln(keys, t + "enc_output_norm", d)
    for i in range(T.ENC_LAYERS):
        p = t + "encoder.layers.%d." % i
        # NB: the reference reads enc/dec n_points from swapped keys (detection_transformer_wobackbone.py:29-30)
        _msda(keys, p + "self_attn", d, heads, L, T.DEC_N_POINTS)
        _ln(keys, p + "norm1", d)
        _lin(keys, p + "linear1", ffn, d)
        _lin(keys, p + "linear2", d, ffn)
        _ln(keys, p + "norm2", d)
    for i in range(T.DEC_LAYERS):
        p = t + "decoder.layers.%d." % i
        _mha(keys, p + "attn_intra", d)
        _ln(keys, p + "norm_intra", d)
        _mha(keys, p + "attn_inter", d)
        _ln(keys, p + "norm_inter", d)
        _msda(keys, p + "attn_cross", d, heads, L, T.ENC_N_POINTS)
        _ln(keys, p + "norm_cross", d)
        _lin(keys, p + "linear1", ffn, d)
        _lin(keys, p + "linear2", d, ffn)
        _ln(keys, p + "norm3", d)
    _mlp(keys, t + "decoder.ref_point_head", d, d, d, 2)
    # heads (one shared module each; canonical copy is index 0)
    _mlp(keys, prefix + "bezier_proposal_coord", d, d, 8, 3)
    _lin(keys, prefix + "bezier_proposal_class", 1, d)
    _mlp(keys, prefix + "ctrl_point_coord.0", d, d, 2, 3)
    _lin(keys, prefix + "ctrl_point_class.0", 1, d)
    _lin(keys, prefix + "ctrl_point_text.0", T.VOC_SIZE + 1, d)
    if T.BOUNDARY_HEAD:
        _mlp(keys, prefix + "boundary_offset.0", d, d, 4, 3)
    return keys


def _matcher_transformer_keys(keys, name, d, n_enc, n_dec, only_crs):
    for i in range(n_enc):
        p = "%s.encoder.layers.%d." % (name, i)
        _mha(keys, p + "self_attn", d)
        _lin(keys, p + "linear1", d, d)
        _lin(keys, p + "linear2", d, d)
    for i in range(n_dec):
        p = "%s.decoder.layers.%d." % (name, i)
        _mha(keys, p + "multihead_attn", d)
        if not only_crs:
            _lin(keys, p + "linear1", d, d)
            _lin(keys, p + "linear2", d, d)


def _roi_head_keys(cfg, prefix="roi_heads."):
    A = cfg.MODEL.ASSO_HEAD
    T = cfg.MODEL.TRANSFORMER
    F = A.FC_DIM
    keys = {}
    din = T.HIDDEN_DIM * T.NUM_POINTS
    for k in range(A.NUM_FC):
        _lin(keys, prefix + "asso_head.fc%d" % (k + 1), F, din)
        din = F
    if cfg.MODEL.ROI_HEADS.WITH_RESR:
        _lin(keys, prefix + "rescoring_head", 1, T.HIDDEN_DIM)
    assert A.NUM_WEIGHT_LAYERS == 0 and A.NO_POS_EMB and not A.NORM, \
        "only the shipped-config head shape (bare dot-product predictor, no pos-emb, no norm) is built"
    name = cfg.MODEL.ROI_HEADS.NAME
    if name == "LSTMatcher":
        for m in ("long_term_matcher", "short_term_matcher"):
            _matcher_transformer_keys(keys, prefix + m, F, A.NUM_ENCODER_LAYERS, A.NUM_DECODER_LAYERS, False)
    elif name == "SHA_FFN_CRSATTN":
        _matcher_transformer_keys(keys, prefix + "shared_matcher", F, 0, A.NUM_DECODER_LAYERS, True)
    else:
        raise ValueError("unknown MODEL.ROI_HEADS.NAME %r" % name)
    return keys


def canonical_keys(cfg):
    keys = {}
    name = cfg.MODEL.BACKBONE.NAME
    keys.update(_swin_keys(swin_type=cfg.MODEL.SWIN.TYPE) if name == "build_swin_backbone" else _vitae_keys() if name == "build_vitaev2_backbone"
                else _resnet_keys())
    keys.update(_deepsolo_keys(cfg))
    keys.update(_roi_head_keys(cfg))
    return keys


def _rng(name, seed):
    key = (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFFFFFFFFFF
    return np.random.Generator(np.random.Philox(key=key))


def _uniform(name, seed, shape, bound):
    return _rng(name, seed).uniform(-bound, bound, size=shape).astype(np.float32)


def synth_state_dict(cfg, seed=0, cls_bias=None, as_torch=True):
    """Synthetic weights with the statistics of the reference's initialisers, except where those
    are degenerate for inference (zero-initialised last layers, prior-probability biases), which
    get small random values so that every stage of the path does observable work."""
    heads = cfg.MODEL.TRANSFORMER.NHEADS
    vitae = cfg.MODEL.BACKBONE.NAME == "build_vitaev2_backbone"
    sd = {}
    for name, shape in canonical_keys(cfg).items():
        leaf = name.rsplit(".", 1)[-1]
        if ".norm." in name and name.startswith("backbone."):
            if leaf == "weight":
                v = 1.0 + _uniform(name, seed, shape, 0.1)
            elif leaf == "running_var":
                v = 1.0 + _uniform(name, seed, shape, 0.2)
            else:
                v = _uniform(name, seed, shape, 0.05)
        elif leaf == "running_var":                          # BatchNorm2d buffers outside the FrozenBN naming (ViTAE PCM)
            v = 1.0 + _uniform(name, seed, shape, 0.2)
        elif vitae and name.startswith("backbone.") and len(shape) == 4:
            # nn.Conv2d's own default (kaiming_uniform(a=sqrt(5)) = U(+-1/sqrt(fan_in))): ViTAE has no output norms, the
            # hotter initialiser used for the FrozenBN R-50 would grow its activations by 100x per stage
            v = _uniform(name, seed, shape, 1.0 / math.sqrt(shape[1] * shape[2] * shape[3]))
        elif vitae and name.startswith("backbone.") and len(shape) == 2:
            v = _uniform(name, seed, shape, 0.02 * math.sqrt(3.0))      # trunc_normal_(std=.02), vitae_v2.py:196
        elif len(shape) == 4:  # conv: kaiming-uniform over fan_in (keeps post-ReLU scale ~stationary)
            fan_in = shape[1] * shape[2] * shape[3]
            v = _uniform(name, seed, shape, math.sqrt(6.0 / fan_in))
        elif name.endswith("sampling_offsets.bias"):
            th = np.arange(heads, dtype=np.float32) * (2.0 * math.pi / heads)
            g = np.stack([np.cos(th), np.sin(th)], -1)
            g = g / np.abs(g).max(-1, keepdims=True)
            npts = shape[0] // (heads * 4 * 2)
            g = np.tile(g.reshape(heads, 1, 1, 2), (1, 4, npts, 1))
            for i in range(npts):
                g[:, :, i, :] *= i + 1
            v = g.reshape(-1).astype(np.float32)
        elif name.endswith("sampling_offsets.weight"):
            v = _uniform(name, seed, shape, 0.02)
        elif name.endswith("attention_weights.weight"):
            v = _uniform(name, seed, shape, 0.05)
        elif len(shape) == 2:
            if name.endswith("level_embed") or name.endswith("point_embed.weight"):
                v = _rng(name, seed).standard_normal(size=shape).astype(np.float32)
            else:  # xavier-uniform
                v = _uniform(name, seed, shape, math.sqrt(6.0 / (shape[0] + shape[1])))
        elif leaf == "weight":  # LayerNorm / GroupNorm gain
            v = 1.0 + _uniform(name, seed, shape, 0.1)
        else:  # biases
            v = _uniform(name, seed, shape, 0.05)
        sd[name] = v
    # the reference zero-inits the last coord/boundary layers; keep them small so refinement moves a little
    for name in list(sd):
        if (".layers.2." in name) and ("coord" in name or "boundary" in name):
            sd[name] = (sd[name] * 0.1).astype(np.float32)
    if cls_bias is not None:
        for k, b in cls_bias.items():
            sd[k] = np.full(sd[k].shape, b, dtype=np.float32)
    if as_torch:
        import torch
        sd = {k: torch.from_numpy(v) for k, v in sd.items()}
    return sd


def normalize_state_dict(sd):
    """Map a reference-style checkpoint onto the canonical key set."""
    if "model" in sd and isinstance(sd["model"], dict):
        sd = sd["model"]
    out = {}
    for k, v in sd.items():
        if k.startswith("module."):
            k = k[len("module."):]
        if k.startswith("detection_transformer.backbone."):
            k = k.replace("detection_transformer.backbone", "backbone", 1)
        out[k] = v
    alias = {
        "detection_transformer.transformer.bezier_coord_embed.": "detection_transformer.bezier_proposal_coord.",
        "detection_transformer.transformer.bezier_class_embed.": "detection_transformer.bezier_proposal_class.",
        "detection_transformer.transformer.decoder.ctrl_point_coord.0.": "detection_transformer.ctrl_point_coord.0.",
    }
    for k in list(out):
        for a, c in alias.items():
            if k.startswith(a) and (c + k[len(a):]) not in out:
                out[c + k[len(a):]] = out[k]
    return out


def expand_for_reference(sd, num_dec_layers=6):
    """Inverse of the alias resolution: the duplicated keys a reference nn.Module state_dict carries."""
    out = dict(sd)
    p = "detection_transformer."
    for k, v in sd.items():
        for head in ("ctrl_point_coord", "ctrl_point_class", "ctrl_point_text", "boundary_offset"):
            h0 = p + head + ".0."
            if k.startswith(h0):
                for i in range(num_dec_layers):
                    out[p + head + ".%d." % i + k[len(h0):]] = v
                    if head == "ctrl_point_coord":
                        out[p + "transformer.decoder.ctrl_point_coord.%d." % i + k[len(h0):]] = v
        if k.startswith(p + "bezier_proposal_coord."):
            out[p + "transformer.bezier_coord_embed." + k[len(p + "bezier_proposal_coord."):]] = v
        if k.startswith(p + "bezier_proposal_class."):
            out[p + "transformer.bezier_class_embed." + k[len(p + "bezier_proposal_class."):]] = v
    return out
