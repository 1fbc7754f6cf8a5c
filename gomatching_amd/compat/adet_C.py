"""Drop-in for the reference's compiled op module `adet._C`
(/root/reference/third_party/adet/layers/csrc/vision.cpp:52-55; declarations
third_party/adet/layers/csrc/DeformAttn/ms_deform_attn.h:20-61; CUDA entry ms_deform_attn_cuda.cu:19-78).

    from gomatching_amd.compat import adet_C
    adet_C.install()                 # sys.modules["adet._C"] = this module: `from adet import _C` now binds the MI355X op

Same two callables, same argument order, same preconditions and error type as the reference:
  * every tensor contiguous and on the GPU                         (ms_deform_attn_cuda.cu:28-38, AT_ASSERTM -> RuntimeError)
  * batch % min(batch, im2col_step) == 0                            (ms_deform_attn_cuda.cu:50-52)
  * inputs borrowed, output freshly allocated in value's dtype/device, launch on the caller's current stream.
The op is also registered with the dispatcher as `torch.ops.gomatching.ms_deform_attn_forward` ("preferred").
Differences, stated rather than hidden: fp32 only (the reference also dispatches fp64, ms_deform_attn_cuda.cu:64) and the
shipped shape 8 heads x 32 channels x 4 levels x 4 points only -- anything else raises RuntimeError instead of running;
`ms_deform_attn_backward` raises NotImplementedError (DeepSolo is frozen in every shipped config, configs/*.yaml:3)."""
import sys

import torch

from .. import ops as _ops
from ..lib import GomError

_NS = "gomatching"


def _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    for name, t in (("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                    ("sampling_loc", sampling_loc), ("attn_weight", attn_weight)):
        if not t.is_contiguous():
            raise RuntimeError("%s tensor has to be contiguous" % name)            # ms_deform_attn_cuda.cu:28-32
        if not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor" % name)                  # :34-38
    if value.dim() != 4 or sampling_loc.dim() != 6 or attn_weight.dim() != 5:
        raise RuntimeError("expected value [B,S,M,D], sampling_loc [B,Lq,M,L,P,2], attn_weight [B,Lq,M,L,P]")
    batch = value.shape[0]
    step = min(batch, int(im2col_step))
    if step <= 0 or batch % step != 0:
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (batch, step))       # :50-52
    if value.dtype != torch.float32:
        raise RuntimeError("ms_deform_attn_forward on MI355X serves float32 only (got %s)" % value.dtype)


@torch.library.custom_op(_NS + "::ms_deform_attn_forward", mutates_args=())
def _op(value: torch.Tensor, spatial_shapes: torch.Tensor, level_start_index: torch.Tensor, sampling_loc: torch.Tensor,
        attn_weight: torch.Tensor, im2col_step: int) -> torch.Tensor:
    _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    try:
        return _ops.ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    except GomError as e:                                        # the reference's AT_ASSERTM / AT_ERROR surface as RuntimeError
        raise RuntimeError(str(e)) from e


@_op.register_fake
def _(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    B, _, M, D = value.shape
    return value.new_empty((B, sampling_loc.shape[1], M * D))


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    return torch.ops.gomatching.ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                                       int(im2col_step))


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    raise NotImplementedError("ms_deform_attn_backward: DeepSolo is frozen on the GoMatching path (configs/*.yaml:3); the "
                              "MI355X library ships the forward op only")


def install(name="adet._C"):
    """Make `from adet import _C` (third_party/adet/layers/ms_deform_attn.py:17) resolve to this module."""
    mod = sys.modules[__name__]
    sys.modules[name] = mod
    parent = sys.modules.get(name.rsplit(".", 1)[0])
    if parent is not None:
        setattr(parent, name.rsplit(".", 1)[1], mod)
    return mod
