"""The META_ARCH class Detectron2 builds for `MODEL.META_ARCHITECTURE: "GoMatchingMI355X"`
(counterpart of /root/reference/gomatching/modeling/meta_arch/gom_lstmatcher.py:113-190).

    from gomatching_amd.compat import d2_register
    d2_register.register()                    # into detectron2's META_ARCH_REGISTRY (or a registry handed in)
    cfg.MODEL.META_ARCHITECTURE = "GoMatchingMI355X"
    model = build_model(cfg); DetectionCheckpointer(model).load(cfg.MODEL.WEIGHTS)      # train_net.py:183-190, eval.py

A real `nn.Module`: every weight of the reference's state dict is a parameter / buffer under the reference's own name, so
the call sites that walk the module tree keep working unchanged --
  * `model.roi_heads.rescoring_head.parameters()` / `model.detection_transformer.ctrl_point_class[-1].parameters()`
    (train_net.py:97-104; the six head copies are ONE shared module, detection_transformer_wobackbone.py:141-153),
  * `model.roi_heads.children()`, `model.parameters()` (gomatching/modeling/freeze_layers.py:20-37),
  * `DetectionCheckpointer.load` -> `load_state_dict` (keys = the reference's; official DeepSolo checkpoints are renamed by
    `weights.normalize_state_dict` with the rule of tools/decouple_deepsolo.py:13-19).
Inference calls (`batch_inference`, `inference`, `_remove_short_track`, `batch_postprocess`, `run_*_match`, `min_track_len`)
go to the HIP implementation, which is (re)built from the CURRENT parameter values whenever they changed since the last
build.  `forward(batched_inputs)` is the reference's training entry (gom_lstmatcher.py:213-266): losses of the trainable
head (`gomatching_amd/training.py`)."""
import torch
from torch import nn

from ..config import CfgNode, _wrap
from ..weights import canonical_keys, normalize_state_dict

ARCH_NAME = "GoMatchingMI355X"
_DELEGATED = ("batch_inference", "inference", "_remove_short_track", "batch_postprocess", "run_short_term_match",
              "run_long_term_match", "detect_launch", "detect_finish", "track_frames", "begin_batch", "preprocess_image")


class _Node(nn.Module):
    """A container named like the reference's sub-module; numeric children index like an nn.ModuleList."""

    def __getitem__(self, i):
        return list(self._modules.values())[i]                  # (children() would drop the shared copies)

    def __len__(self):
        return len(self._modules)


def _is_buffer(key):
    # Detectron2's FrozenBatchNorm2d keeps weight / bias / running stats as buffers; so do the BatchNorm statistics of ViTAE
    return ".norm." in key and key.startswith("backbone.") or key.endswith(("running_mean", "running_var", "num_batches_tracked"))


def _cfg_of(cfg):
    if isinstance(cfg, CfgNode):
        return cfg
    if hasattr(cfg, "dump") and hasattr(cfg, "items"):          # a yacs / Detectron2 CfgNode
        import yaml
        return _wrap(yaml.safe_load(cfg.dump()))
    return _wrap(dict(cfg))


class GoMatchingMI355X(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = _cfg_of(cfg)
        V = self.cfg.VIDEO_TEST
        self.min_track_len = V.MIN_TRACK_LEN
        self.with_rescore = self.cfg.MODEL.ROI_HEADS.WITH_RESR
        n_dec = self.cfg.MODEL.TRANSFORMER.DEC_LAYERS
        shared = {}
        for key, shape in canonical_keys(self.cfg).items():
            parts = key.split(".")
            node = self
            for depth, p in enumerate(parts[:-1]):
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
                # the per-decoder-layer head copies of DeepSolo are one shared module listed n_dec times
                if parts[0] == "detection_transformer" and depth == 1 and parts[1] in (
                        "ctrl_point_class", "ctrl_point_coord", "ctrl_point_text", "boundary_offset") and parts[2] == "0":
                    shared[parts[1]] = True
            t = torch.zeros(tuple(shape), dtype=torch.float32)
            if _is_buffer(key):
                node.register_buffer(parts[-1], t)
            else:
                node.register_parameter(parts[-1], nn.Parameter(t, requires_grad=parts[0] == "roi_heads"))
        dt = self._modules.get("detection_transformer")
        for name in shared:
            lst = dt._modules[name]
            for i in range(1, n_dec):
                if str(i) not in lst._modules:
                    lst.add_module(str(i), lst._modules["0"])        # same object: parameters() yields it once
        self._impl = None
        self._impl_version = None
        self._device = torch.device(self.cfg.MODEL.DEVICE if self.cfg.MODEL.DEVICE != "cpu" else "cpu")

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict, strict=False, assign=False):
        """Accepts the reference's checkpoints (incl. un-decoupled DeepSolo ones and the six expanded head copies)."""
        sd = normalize_state_dict(dict(state_dict))
        own = dict(self.named_parameters(remove_duplicate=True))
        own.update(dict(self.named_buffers()))
        missing, unexpected = [], []
        with torch.no_grad():
            for k, dst in own.items():
                if k in sd:
                    src = torch.as_tensor(sd[k]).to(dst.dtype)
                    if tuple(src.shape) != tuple(dst.shape):
                        raise RuntimeError("size mismatch for %s: checkpoint %s vs model %s" % (k, tuple(src.shape), tuple(dst.shape)))
                    dst.copy_(src)
                else:
                    missing.append(k)
        from ..weights import expand_for_reference
        aliases = set(expand_for_reference({k: None for k in own}, self.cfg.MODEL.TRANSFORMER.DEC_LAYERS))
        unexpected = [k for k in sd if k not in own and k not in aliases]       # the reference's duplicated head keys are fine
        if strict and (missing or unexpected):
            raise RuntimeError("missing keys %s, unexpected keys %s" % (missing[:5], unexpected[:5]))
        self._impl = None
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def _version(self, frozen_only=False):
        return tuple(p._version for k, p in self.named_parameters() if not (frozen_only and k.startswith("roi_heads."))) \
            + tuple(b._version for _, b in self.named_buffers()) + (str(self._device),)

    def impl(self, for_training=False):
        """The HIP model, rebuilt when a parameter changed (optimizer step, checkpoint load, .to()).  `for_training`: the caller
        (training.forward_losses) uses only the FROZEN detector of the HIP model and reads the head from the live parameters, so
        an optimizer step -- which bumps the versions of `roi_heads.*` only (freeze_layers.py:20-37) -- must not re-prepare the
        backbone and DeepSolo every iteration; the first inference call after training sees the head's new versions and rebuilds."""
        if for_training and self._impl is not None and getattr(self, "_impl_frozen_version", None) == self._version(True):
            return self._impl
        v = self._version()
        if self._impl is None or self._impl_version != v:
            from ..modeling import GoMatching
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("GoMatchingMI355X runs on an MI355X only: move the model to the GPU first (model.to('cuda'))")
            sd = {k: t.detach() for k, t in self.state_dict().items()}
            self._impl = GoMatching(self.cfg, sd, device=dev)
            self._impl_version = v
            self._impl_frozen_version = self._version(True)
        return self._impl

    def __getattr__(self, name):
        if name in _DELEGATED:
            return getattr(self.impl(), name)
        return super().__getattr__(name)

    @property
    def device(self):
        return next(self.parameters()).device

    # ------------------------------------------------------------------ training entry
    def forward(self, batched_inputs):
        if not self.training:
            raise RuntimeError("inference goes through batch_inference / inference (gom_lstmatcher.py:268,366), as in the "
                               "reference; forward() is the training entry")
        from .. import training
        return training.forward_losses(self, batched_inputs)


class _RoiHeadsMI355X(nn.Module):
    """The association head as the class `build_roi_heads(cfg, input_shape)` constructs for `MODEL.ROI_HEADS.NAME`
    (/root/reference/gomatching/modeling/roi_heads/lstmatcher.py:59-92, shared_ffn_crsattn.py:62-63): every `roi_heads.*` weight of
    the reference's state dict is a parameter under the reference's name (prefix stripped), so `load_state_dict`, `children()`
    and `rescoring_head.parameters()` work; the eval-time arithmetic (`match_scores`, `short_term_scores`, `_forward_transformer`,
    `_activate_asso`; `impl().asso_head` / `impl().rescoring_head`, whose names are sub-modules here) goes to the HIP head (`modeling/roi_heads.py`), rebuilt from the
    CURRENT parameter values when they changed."""
    HEAD_NAME = None
    _HIP = ("match_scores", "short_term_scores", "_forward_transformer", "_activate_asso", "feature_dim",
            "asso_thresh_test")

    def __init__(self, cfg, input_shape=None):
        super().__init__()
        self.cfg = _cfg_of(cfg).clone()
        self.cfg.MODEL.ROI_HEADS.NAME = self.HEAD_NAME
        for key, shape in canonical_keys(self.cfg).items():
            parts = key.split(".")
            if parts[0] != "roi_heads":
                continue
            node = self
            for p in parts[1:-1]:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            node.register_parameter(parts[-1], nn.Parameter(torch.zeros(tuple(shape), dtype=torch.float32)))
        self._hip = None
        self._hip_version = None

    def impl(self):
        v = tuple(p._version for p in self.parameters()) + (str(next(self.parameters()).device),)
        if self._hip is None or self._hip_version != v:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("%s runs on an MI355X only: move the module to the GPU first" % type(self).__name__)
            from ..modeling.roi_heads import build_roi_heads
            self._hip = build_roi_heads(self.cfg, {"roi_heads." + k: t.detach() for k, t in self.state_dict().items()}, dev)
            self._hip_version = v
        return self._hip

    def __getattr__(self, name):
        if name in type(self)._HIP:
            return getattr(self.impl(), name)
        return super().__getattr__(name)

    def forward(self, *args, **kw):
        raise RuntimeError("the head is driven by the META_ARCH (gom_lstmatcher.py:157,286-349): build MODEL.META_ARCHITECTURE "
                           "'GoMatching' / '%s' from gomatching_amd.compat.d2_register" % ARCH_NAME)


class LSTMatcher(_RoiHeadsMI355X):
    HEAD_NAME = "LSTMatcher"


class SHA_FFN_CRSATTN(_RoiHeadsMI355X):
    HEAD_NAME = "SHA_FFN_CRSATTN"


_NAMED = {}


def register(registry=None, name=None, roi_heads_registry=None):
    """Register the META_ARCH class in Detectron2's META_ARCH_REGISTRY (or in `registry`, any object with the fvcore Registry
    interface) under `name` (default ARCH_NAME).  `name="GoMatching"` takes the REFERENCE's name, so its yaml files
    (`MODEL.META_ARCHITECTURE: "GoMatching"`, configs/*.yaml:2) build the MI355X model unchanged -- for checkouts that do not
    import `gomatching.modeling` (whose own registration of that name would collide).  `roi_heads_registry` (or True for
    Detectron2's ROI_HEADS_REGISTRY) additionally offers `LSTMatcher` / `SHA_FFN_CRSATTN` under the reference's names
    (lstmatcher.py:59-60, shared_ffn_crsattn.py:62-63) to `build_roi_heads`.  Returns the registered META_ARCH class."""
    if registry is None:
        from detectron2.modeling.meta_arch.build import META_ARCH_REGISTRY as registry
    cls = GoMatchingMI355X
    if name is not None and name != ARCH_NAME:
        cls = _NAMED.get(str(name))
        if cls is None:
            cls = _NAMED[str(name)] = type(str(name), (GoMatchingMI355X,), {"__doc__": GoMatchingMI355X.__doc__,
                                                                            "__module__": __name__})
    try:
        present = registry.get(cls.__name__)
    except (KeyError, AssertionError):
        present = None
    if present is None:
        registry.register(cls)
    elif present is not cls and not (isinstance(present, type) and issubclass(present, GoMatchingMI355X)):
        raise RuntimeError("META_ARCH name %r is already taken by %r (importing gomatching.modeling registers the reference's "
                           "class under it): register the MI355X model under %r instead" % (cls.__name__, present, ARCH_NAME))
    else:
        cls = present
    if roi_heads_registry is not None and roi_heads_registry is not False:
        if roi_heads_registry is True:
            from detectron2.modeling.roi_heads.roi_heads import ROI_HEADS_REGISTRY as roi_heads_registry
        for head in (LSTMatcher, SHA_FFN_CRSATTN):
            try:
                roi_heads_registry.register(head)
            except (AssertionError, KeyError):
                pass
    return cls
