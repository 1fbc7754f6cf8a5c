"""CPU: the shipped binding stubs (gomatching_amd/compat) -- the op module's preconditions / dispatcher registration and the
META_ARCH wrapper built against the shim registry of oracle/ref_shim.py, loaded from a synthetic checkpoint, serving the
attribute paths the reference's training script walks (train_net.py:97-104, freeze_layers.py:20-37)."""
import pytest
import torch

from helpers import mini_cfg
from gomatching_amd.weights import synth_state_dict, expand_for_reference


def test_adet_C_preconditions_and_dispatcher_registration():
    from gomatching_amd.compat import adet_C
    assert hasattr(torch.ops.gomatching, "ms_deform_attn_forward")
    v = torch.zeros(2, 10, 8, 32)
    shapes, lsi = torch.zeros(4, 2, dtype=torch.long), torch.zeros(4, dtype=torch.long)
    loc, w = torch.zeros(2, 5, 8, 4, 4, 2), torch.zeros(2, 5, 8, 4, 4)
    with pytest.raises(RuntimeError, match="CUDA"):              # ms_deform_attn_cuda.cu:34-38
        adet_C.ms_deform_attn_forward(v, shapes, lsi, loc, w, 64)
    with pytest.raises(RuntimeError, match="contiguous"):        # :28-32
        adet_C.ms_deform_attn_forward(v.transpose(1, 2), shapes, lsi, loc, w, 64)
    with pytest.raises(RuntimeError, match="CUDA"):              # the backward asserts the same (:92-106)
        adet_C.ms_deform_attn_backward(v, shapes, lsi, loc, w, torch.zeros(2, 5, 256), 64)
    assert hasattr(torch.ops.gomatching, "ms_deform_attn_backward")
    gv, gl, gw = torch.ops.gomatching.ms_deform_attn_backward(*(x.to("meta") for x in (v, shapes, lsi, loc, w,
                                                                                        torch.zeros(2, 5, 256))), 64)
    assert gv.shape == v.shape and gl.shape == loc.shape and gw.shape == w.shape
    # shape inference through the dispatcher (meta tensors): [B, Lq, M*D]
    out = torch.ops.gomatching.ms_deform_attn_forward(v.to("meta"), shapes.to("meta"), lsi.to("meta"), loc.to("meta"),
                                                      w.to("meta"), 64)
    assert tuple(out.shape) == (2, 5, 256)
    import sys
    adet_C.install("adet_test_pkg._C")
    assert sys.modules["adet_test_pkg._C"].ms_deform_attn_forward is adet_C.ms_deform_attn_forward


def test_meta_arch_wrapper_in_the_shim_registry():
    from oracle import ref_shim
    from gomatching_amd.compat import d2_register
    ref_shim.install()
    from detectron2.modeling.meta_arch.build import META_ARCH_REGISTRY
    d2_register.register(META_ARCH_REGISTRY)
    cls = META_ARCH_REGISTRY.get(d2_register.ARCH_NAME)
    cfg = mini_cfg("icdar15")
    model = cls(cfg)
    assert isinstance(model, torch.nn.Module)
    sd = synth_state_dict(cfg, seed=3)
    res = model.load_state_dict(expand_for_reference(sd))        # a reference checkpoint carries the six head copies
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in sd.items():
        assert torch.equal(model.state_dict()[k], torch.as_tensor(v).float()), k
    # train_net.py:97-104: rescoring head initialised from the (shared) last classifier copy
    heads = model.detection_transformer.ctrl_point_class
    assert len(heads) == cfg.MODEL.TRANSFORMER.DEC_LAYERS and heads[-1] is heads[0]
    for pk, pq in zip(model.roi_heads.rescoring_head.parameters(), heads[-1].parameters()):
        pk.data = pq.data.clone().detach()
    assert torch.equal(model.roi_heads.rescoring_head.weight, torch.as_tensor(sd["detection_transformer.ctrl_point_class.0.weight"]))
    # freeze_layers.py:20-37: only roi_heads trains; parameter counts of README / SURVEY (32.79 M for LSTMatcher)
    names = [n for n, _ in model.roi_heads.named_children()]
    assert names == ["asso_head", "rescoring_head", "long_term_matcher", "short_term_matcher"]
    trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert trainable == 32794881
    assert all(not p.requires_grad for n, p in model.named_parameters() if not n.startswith("roi_heads."))
    assert model.min_track_len == cfg.VIDEO_TEST.MIN_TRACK_LEN
    with pytest.raises(RuntimeError, match="MI355X"):            # no CPU path: inference before .to('cuda') fails loudly
        model.batch_inference([], 0, 0, [], {})


def test_registration_under_the_reference_names():
    """north_star: the reference's yaml files name `MODEL.META_ARCHITECTURE: "GoMatching"` (configs/*.yaml:2) and
    `MODEL.ROI_HEADS.NAME: LSTMatcher | SHA_FFN_CRSATTN`; `register(name="GoMatching", roi_heads_registry=...)` offers the MI355X
    classes under exactly those names (for checkouts that do not import `gomatching.modeling`, whose registration would collide)."""
    from oracle import ref_shim
    from gomatching_amd.compat import d2_register
    from gomatching_amd.weights import canonical_keys
    ref_shim.install()
    meta, heads = type(ref_shim._Registry())(), type(ref_shim._Registry())()
    cls = d2_register.register(meta, name="GoMatching", roi_heads_registry=heads)
    assert meta.get("GoMatching") is cls and cls.__name__ == "GoMatching" and issubclass(cls, d2_register.GoMatchingMI355X)
    assert d2_register.register(meta, name="GoMatching") is cls                 # idempotent
    d2_register.register(meta)                                                  # and the MI355X name beside it
    assert meta.get(d2_register.ARCH_NAME) is d2_register.GoMatchingMI355X
    for builtin, name, n_params in (("icdar15", "LSTMatcher", 32794881), ("pp_dstext", "SHA_FFN_CRSATTN", 11802624)):
        cfg = mini_cfg(builtin)
        assert cfg.MODEL.ROI_HEADS.NAME == name
        head = heads.get(name)(cfg, None)                                        # build_roi_heads(cfg, input_shape)
        assert isinstance(head, torch.nn.Module) and type(head).__name__ == name
        want = {k[len("roi_heads."):]: tuple(s) for k, s in canonical_keys(cfg).items() if k.startswith("roi_heads.")}
        got = {k: tuple(p.shape) for k, p in head.named_parameters()}
        assert got == want
        n = sum(p.numel() for p in head.parameters())
        assert n == n_params, (name, n)                                        # (pp_dstext: no rescoring head, 257 fewer)
        sd = synth_state_dict(cfg, seed=3)
        res = head.load_state_dict({k[len("roi_heads."):]: torch.as_tensor(v).float() for k, v in sd.items()
                                    if k.startswith("roi_heads.")})
        assert not res.missing_keys and not res.unexpected_keys
        with pytest.raises(RuntimeError, match="MI355X"):
            head.match_scores                                                    # the arithmetic lives on the GPU only
        model = cls(cfg)                                                         # the yaml's META_ARCHITECTURE builds the wrapper
        assert [n_ for n_, _ in model.roi_heads.named_children()][0] == "asso_head"
