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
float32 and float64 (the reference's AT_DISPATCH_FLOATING_TYPES, ms_deform_attn_cuda.cu:64), any heads / channels / levels /
points: the shipped shape (8 x 32 x 4 x 4, fp32) runs msda.hip's wave-per-query kernel, everything else and the backward the
general kernels of msda_any.hip.  `ms_deform_attn_backward` returns [grad_value, grad_sampling_loc, grad_attn_weight] as the
reference's does (ms_deform_attn_cuda.cu:83-156), so the reference's own `MSDeformAttnFunction` (ms_deform_attn.py:20-37)
trains through this module unchanged; the dispatcher op carries the same backward for `torch.ops` callers."""
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
    if value.dtype not in (torch.float32, torch.float64):              # the dispatch macro's own refusal (:64)
        raise RuntimeError('"ms_deform_attn_forward_cuda" not implemented for \'%s\'' % str(value.dtype).replace("torch.", ""))


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


@torch.library.custom_op(_NS + "::ms_deform_attn_backward", mutates_args=())
def _bwd_op(value: torch.Tensor, spatial_shapes: torch.Tensor, level_start_index: torch.Tensor, sampling_loc: torch.Tensor,
            attn_weight: torch.Tensor, grad_output: torch.Tensor, im2col_step: int) -> list[torch.Tensor]:
    _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    if not grad_output.is_contiguous():
        raise RuntimeError("grad_output tensor has to be contiguous")              # ms_deform_attn_cuda.cu:99
    if not grad_output.is_cuda:
        raise RuntimeError("grad_output must be a CUDA tensor")                    # :106
    try:
        return list(_ops.ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                                 grad_output, im2col_step))
    except GomError as e:
        raise RuntimeError(str(e)) from e


@_bwd_op.register_fake
def _(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    return [torch.empty_like(value), torch.empty_like(sampling_loc), torch.empty_like(attn_weight)]


def _setup_context(ctx, inputs, output):
    ctx.save_for_backward(*inputs[:5])
    ctx.im2col_step = inputs[5]


def _backward(ctx, grad_output):
    value, shapes, lsi, loc, w = ctx.saved_tensors
    gv, gl, gw = torch.ops.gomatching.ms_deform_attn_backward(value, shapes, lsi, loc, w, grad_output.contiguous(),
                                                              ctx.im2col_step)
    return gv, None, None, gl, gw, None


_op.register_autograd(_backward, setup_context=_setup_context)


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    return torch.ops.gomatching.ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                                        grad_output, int(im2col_step))


def install(name="adet._C"):
    """Make `from adet import _C` (third_party/adet/layers/ms_deform_attn.py:17) resolve to this module."""
    mod = sys.modules[__name__]
    sys.modules[name] = mod
    parent = sys.modules.get(name.rsplit(".", 1)[0])
    if parent is not None:
        setattr(parent, name.rsplit(".", 1)[1], mod)
    return mod
