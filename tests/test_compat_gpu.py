"""GPU: the shipped binding stubs end to end -- the `adet._C`-shaped op on the reference's own fixture, and the META_ARCH
wrapper (nn.Module, weights loaded through load_state_dict) reproducing the reference's track ids."""
import pytest
import torch

from helpers import mini_cfg, golden, e2e_state_dict, t

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("case", ["enc", "dec", "oob"])
def test_adet_C_module_on_the_reference_fixture(case):
    from gomatching_amd.compat import adet_C
    g = golden("msda.npz")
    args = [t(g[case + k]).to(DEV) for k in ("_value", "_shapes", "_lsi", "_loc", "_w")]
    out = adet_C.ms_deform_attn_forward(*args, 64)
    assert out.dtype == torch.float32 and out.is_cuda
    assert float((out.cpu() - t(g[case + "_out"])).abs().max()) <= 2e-5
    assert torch.equal(out, torch.ops.gomatching.ms_deform_attn_forward(*args, 64))
    # batch % min(batch, im2col_step) != 0 (ms_deform_attn_cuda.cu:50-52): three copies of the first image, step 2
    three = [args[0][:1].repeat(3, 1, 1, 1).contiguous(), args[1], args[2],
             args[3][:1].repeat(3, 1, 1, 1, 1, 1).contiguous(), args[4][:1].repeat(3, 1, 1, 1, 1).contiguous()]
    with pytest.raises(RuntimeError, match="im2col_step"):
        adet_C.ms_deform_attn_forward(*three, 2)
    assert torch.equal(adet_C.ms_deform_attn_forward(*three, 3)[0], out[0])
    # the reference dispatches on the dtype (ms_deform_attn_cuda.cu:64): float64 runs, half is refused by the macro
    out64 = adet_C.ms_deform_attn_forward(args[0].double(), args[1], args[2], args[3].double(), args[4].double(), 64)
    assert out64.dtype == torch.float64 and float((out64.cpu() - t(g[case + "_out"]).double()).abs().max()) <= 2e-6
    with pytest.raises(RuntimeError, match="not implemented for 'float16'"):
        adet_C.ms_deform_attn_forward(args[0].half(), args[1], args[2], args[3].half(), args[4].half(), 64)


def test_meta_arch_wrapper_reproduces_reference_ids():
    from gomatching_amd.compat.d2_register import GoMatchingMI355X
    from gomatching_amd.synth import make_clip
    from gomatching_amd.weights import expand_for_reference
    g = golden("e2e_lst.npz")
    cfg = mini_cfg("icdar15", device="cuda")
    model = GoMatchingMI355X(cfg).to(DEV).eval()
    res = model.load_state_dict(expand_for_reference(e2e_state_dict(cfg, g)))
    assert not res.missing_keys and not res.unexpected_keys
    hw = tuple(int(v) for v in g["hw"])
    clip = make_clip(8, hw[0], hw[1], clip_id=1)
    inputs = [{"image": torch.as_tensor(f.astype("float32").transpose(2, 0, 1))} for f in clip]
    tc = {k: 0.0 for k in ("pre_process", "backbone", "detector", "rescore", "tracker", "short_match", "long_match")}
    insts, id_count = model.batch_inference(inputs, 0, 0, [], tc)
    assert [x.track_ids.cpu().tolist() for x in insts] == [g["pre_ids_%d" % f].tolist() for f in range(8)]
    impl = model.impl()
    assert model.impl() is impl                                   # not rebuilt while the parameters are unchanged
    with torch.no_grad():
        model.roi_heads.rescoring_head.bias.add_(1.0)             # an optimizer step would do this
    assert model.impl(for_training=True) is impl                  # the training entry only needs the FROZEN detector: kept
    new = model.impl()                                            # inference sees the head's new version: rebuilt
    assert new is not impl
    with torch.no_grad():
        next(p for k, p in model.named_parameters() if not k.startswith("roi_heads.")).add_(0.0)   # a frozen weight touched
    assert model.impl(for_training=True) is not new               # ... now the training entry rebuilds as well
