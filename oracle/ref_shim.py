"""Import shim for the *reference* GoMatching sources (container-only tooling).

TEST INFRASTRUCTURE -- never imported by the product path (`gomatching_amd/`).

`/root/reference` needs Detectron2 v0.6, torchvision, fvcore and the compiled
`adet._C` CUDA op, none of which exist in this image.  This module installs
`sys.modules` stand-ins for exactly the third-party *symbols* the inference
path touches, so the reference's own arithmetic files
import and execute unmodified on CPU.  It is used ONLY by
`oracle/gen_golden.py` to produce the committed fixtures in `tests/golden/`;
nothing here (and nothing under /root/reference) travels to the GPU box.

The Detectron2 containers (`Instances`, `Boxes`, `pairwise_iou`, `nms`,
`ImageList`) are restated from their published v0.6 semantics: they are NOT
reference code, so parity at those call sites is "unpinned" (DESIGN.md).
"""
import importlib
import os
import sys
import types

import torch
from torch import nn

REF_ROOT = os.environ.get("GOM_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "gomatching"))


# --------------------------------------------------------------------------
# Detectron2-like containers (restated from published semantics)
# --------------------------------------------------------------------------
class Boxes:
    def __init__(self, tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        tensor = tensor.to(torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4))
        self.tensor = tensor

    def to(self, device):
        return Boxes(self.tensor.to(device))

    def clone(self):
        return Boxes(self.tensor.clone())

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        return Boxes(self.tensor[item])

    def __len__(self):
        return self.tensor.shape[0]

    @property
    def device(self):
        return self.tensor.device


def pairwise_iou(boxes1, boxes2):
    a, b = boxes1.tensor, boxes2.tensor
    area1, area2 = boxes1.area(), boxes2.area()
    wh = torch.min(a[:, None, 2:], b[:, 2:]) - torch.max(a[:, None, :2], b[:, :2])
    wh.clamp_(min=0)
    inter = wh.prod(dim=2)
    iou = torch.where(
        inter > 0,
        inter / (area1[:, None] + area2 - inter),
        torch.zeros(1, dtype=inter.dtype, device=inter.device),
    )
    return iou


def nms(boxes, scores, iou_threshold):
    """Greedy NMS, indices returned in decreasing score order (torchvision semantics)."""
    import numpy as np
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order].detach().cpu().numpy().astype(np.float32)
    n = b.shape[0]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    zero = np.float32(0)
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(int(order[i]))
        if i + 1 >= n:
            break
        r = b[i + 1:]
        w = np.maximum(zero, np.minimum(b[i, 2], r[:, 2]) - np.maximum(b[i, 0], r[:, 0]))
        h = np.maximum(zero, np.minimum(b[i, 3], r[:, 3]) - np.maximum(b[i, 1], r[:, 1]))
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (area[i] + area[i + 1:] - inter)
        suppressed[i + 1:] |= ovr > np.float32(iou_threshold)
    return torch.as_tensor(keep, dtype=torch.long)


class Instances:
    def __init__(self, image_size, **kwargs):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            object.__setattr__(self, name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))
        return self._fields[name]

    def set(self, name, value):
        data_len = len(value)
        if len(self._fields):
            assert len(self) == data_len, \
                "Adding a field of length {} to a Instances of length {}".format(data_len, len(self))
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def remove(self, name):
        del self._fields[name]

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def to(self, *args, **kwargs):
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            if hasattr(v, "to"):
                v = v.to(*args, **kwargs)
            ret.set(k, v)
        return ret

    def __getitem__(self, item):
        if isinstance(item, int):
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self):
        for v in self._fields.values():
            return v.__len__()
        raise NotImplementedError("Empty Instances does not support __len__!")


class ImageList:
    def __init__(self, tensor, image_sizes):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        image_sizes = [(int(t.shape[-2]), int(t.shape[-1])) for t in tensors]
        max_h = max(s[0] for s in image_sizes)
        max_w = max(s[1] for s in image_sizes)
        if size_divisibility > 1:
            st = size_divisibility
            max_h = (max_h + st - 1) // st * st
            max_w = (max_w + st - 1) // st * st
        out = tensors[0].new_full((len(tensors), tensors[0].shape[0], max_h, max_w), pad_value)
        for i, t in enumerate(tensors):
            out[i, :, : t.shape[-2], : t.shape[-1]].copy_(t)
        return ImageList(out.contiguous(), image_sizes)


class _Registry(dict):
    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self[o.__name__] = o
                return o
            return deco
        self[obj.__name__] = obj
        return obj


def configurable(init_func=None, *, from_config=None):
    """Minimal `detectron2.config.configurable`: cls(cfg, ...) -> cls(**cls.from_config(cfg, ...))."""
    import functools

    assert init_func is not None and init_func.__name__ == "__init__"

    @functools.wraps(init_func)
    def wrapped(self, *args, **kwargs):
        first = args[0] if args else kwargs.get("cfg", None)
        if first is not None and hasattr(first, "MODEL") and hasattr(type(self), "from_config"):
            explicit = type(self).from_config(*args, **kwargs)
            init_func(self, **explicit)
        else:
            init_func(self, *args, **kwargs)

    return wrapped


class CfgNode(dict):
    """Attribute dict standing in for yacs CfgNode (read-only use by the reference)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def to_cfgnode(d):
    if isinstance(d, dict):
        return CfgNode({k: to_cfgnode(v) for k, v in d.items()})
    return d


_installed = False


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def install():
    """Install the stand-ins and the namespace packages that skip the reference __init__ files."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    R = REF_ROOT + "/"

    # --- third-party symbols --------------------------------------------
    def box_area(b):
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    _mod("torchvision")
    _mod("torchvision.ops")
    _mod("torchvision.ops.boxes", box_area=box_area)

    def c2_xavier_fill(module):
        nn.init.kaiming_uniform_(module.weight, a=1)
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)

    _mod("fvcore")
    _mod("fvcore.nn")
    wi = _mod("fvcore.nn.weight_init", c2_xavier_fill=c2_xavier_fill)
    sys.modules["fvcore.nn"].weight_init = wi

    class ShapeSpec:
        def __init__(self, channels=None, height=None, width=None, stride=None):
            self.channels, self.height, self.width, self.stride = channels, height, width, stride

    META = _Registry()
    ROI = _Registry()

    class ROIHeads(nn.Module):
        pass

    def _unused(*a, **k):
        raise RuntimeError("training-only Detectron2 symbol called on the inference path")

    class _Meta:
        @staticmethod
        def get(name):
            return types.SimpleNamespace(name=name)

    _mod("detectron2")
    _mod("detectron2.config", configurable=configurable, CfgNode=CfgNode)
    _mod("detectron2.structures", Boxes=Boxes, pairwise_iou=pairwise_iou, Instances=Instances,
         ImageList=ImageList)
    _mod("detectron2.layers", Linear=nn.Linear, ShapeSpec=ShapeSpec, nms=nms)
    _mod("detectron2.modeling", build_backbone=_unused, build_roi_heads=_unused)
    _mod("detectron2.modeling.meta_arch")
    _mod("detectron2.modeling.meta_arch.build", META_ARCH_REGISTRY=META)
    _mod("detectron2.modeling.roi_heads")
    _mod("detectron2.modeling.roi_heads.roi_heads", ROI_HEADS_REGISTRY=ROI, ROIHeads=ROIHeads)
    _mod("detectron2.modeling.matcher", Matcher=lambda *a, **k: None)
    _mod("detectron2.modeling.sampling", subsample_labels=_unused)
    _mod("detectron2.modeling.proposal_generator")
    _mod("detectron2.modeling.proposal_generator.proposal_utils",
         add_ground_truth_to_proposals=_unused)
    _mod("detectron2.utils")
    _mod("detectron2.utils.events", get_event_storage=_unused)
    _mod("detectron2.utils.comm", get_world_size=lambda: 1)
    _mod("detectron2.data", MetadataCatalog=_Meta)

    # --- namespace packages over the reference tree -----------------------
    _pkg("adet", R + "third_party/adet")
    _pkg("adet.layers", R + "third_party/adet/layers")
    _pkg("adet.modeling", R + "third_party/adet/modeling")
    _pkg("adet.modeling.model", R + "third_party/adet/modeling/model")
    _pkg("adet.utils", R + "third_party/adet/utils")
    _pkg("gomatching", R + "gomatching")
    _pkg("gomatching.modeling", R + "gomatching/modeling")
    _pkg("gomatching.modeling.roi_heads", R + "gomatching/modeling/roi_heads")
    _pkg("gomatching.modeling.meta_arch", R + "gomatching/modeling/meta_arch")

    # --- adet._C -> the reference's own pure-PyTorch core -------------------
    C = _mod("adet._C")
    sys.modules["adet"]._C = C
    msda = importlib.import_module("adet.layers.ms_deform_attn")

    def ms_deform_attn_forward(value, shapes, lsi, loc, w, step):
        return msda.ms_deform_attn_core_pytorch(
            value, [(int(h), int(w_)) for h, w_ in shapes.tolist()], loc, w)

    C.ms_deform_attn_forward = ms_deform_attn_forward
    C.ms_deform_attn_backward = _unused
    _installed = True


def load(name):
    """Import a reference module by dotted name, e.g. 'gomatching.modeling.roi_heads.lstmatcher'."""
    install()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return importlib.import_module(name)
