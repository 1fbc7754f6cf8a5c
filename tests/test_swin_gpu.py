"""Swin-T backbone on the GPU against the outputs of the reference's own SwinTransformer module
(tests/golden/swin_tiny.npz): every stage output within fp32 accumulation noise, for sizes that need the internal window /
patch padding too, under all three contraction back-ends; plus the glue kernels one by one against torch."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import golden
from gomatching_amd.config import setup_cfg
from gomatching_amd.weights import synth_state_dict
from oracle import swin_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(params=["f16x3", "bf16x6", "fp32"])
def gemm_mode(request):
    from gomatching_amd import ops
    old = ops.GEMM_MODE
    ops.GEMM_MODE = request.param
    yield request.param
    ops.GEMM_MODE = old


def _sd():
    cfg = setup_cfg(builtin="icdar15")
    cfg.MODEL.BACKBONE.NAME = "build_swin_backbone"
    return synth_state_dict(cfg, seed=3)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_swin_tiny_matches_reference_module(tag, gemm_mode):
    from gomatching_amd.modeling.swin import SwinTiny
    g = golden("swin_tiny.npz")
    net = SwinTiny(_sd(), torch.device(DEV))
    x = torch.from_numpy(g["x_" + tag])
    x4 = torch.cat([x.permute(0, 2, 3, 1), x.new_zeros(x.shape[0], x.shape[2], x.shape[3], 1)], -1).contiguous().to(DEV)
    out = net.forward(x4)
    for k in ("stage3", "stage4", "stage5"):
        ref = torch.from_numpy(g["%s_%s" % (k, tag)])
        got = out[k].permute(0, 3, 1, 2).cpu()
        assert tuple(got.shape) == tuple(ref.shape)
        err = float((got - ref).abs().max())
        assert err <= 2e-4, (k, tag, err)


def test_swin_small_matches_reference_module(gemm_mode):
    """Swin-S: the same kernels over 18 stage-3 blocks (the other type detection_transformer_wobackbone.py:59-62 admits)."""
    from gomatching_amd.modeling.swin import SwinTiny
    cfg = setup_cfg(builtin="icdar15")
    cfg.MODEL.BACKBONE.NAME = "build_swin_backbone"
    cfg.MODEL.SWIN.TYPE = "small"
    g = golden("swin_small.npz")
    net = SwinTiny(synth_state_dict(cfg, seed=4), torch.device(DEV), swin_type="small")
    assert len(net.stages[2]["blocks"]) == 18
    x = torch.from_numpy(g["x_s"])
    x4 = torch.cat([x.permute(0, 2, 3, 1), x.new_zeros(x.shape[0], x.shape[2], x.shape[3], 1)], -1).contiguous().to(DEV)
    out = net.forward(x4)
    for k in ("stage3", "stage4", "stage5"):
        err = float((out[k].permute(0, 3, 1, 2).cpu() - torch.from_numpy(g[k + "_s"])).abs().max())
        assert err <= 3e-4, (k, err)


def test_swin_glue_kernels_vs_torch():
    from gomatching_amd import ops
    g = torch.Generator().manual_seed(0)
    for D in (96, 192, 384, 768, 1536):
        x = torch.randn(37, D, generator=g) * 3 + 0.7
        gm, bt = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g)
        out = ops.layernorm_any(x.to(DEV), gm.to(DEV), bt.to(DEV)).cpu()
        assert float((out - F.layer_norm(x, (D,), gm, bt)).abs().max()) <= 2e-5
    x = torch.randn(1000, generator=g) * 3
    assert float((ops.gelu_(x.clone().to(DEV)).cpu() - F.gelu(x)).abs().max()) <= 1e-6
    B, H, W, C = 2, 12, 17, 96
    t = torch.randn(B, H, W, C, generator=g)
    for shift in (0, 3):
        win = ops.swin_window_gather(t.view(-1, C).to(DEV), B, H, W, shift).cpu()
        xp = F.pad(t, (0, 0, 0, (7 - W % 7) % 7, 0, (7 - H % 7) % 7))
        Hp, Wp = xp.shape[1], xp.shape[2]
        if shift:
            xp = torch.roll(xp, shifts=(-shift, -shift), dims=(1, 2))
        ref = xp.view(B, Hp // 7, 7, Wp // 7, 7, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, C)
        assert torch.equal(win, ref)
        back = ops.swin_window_scatter_add(win.to(DEV), t.view(-1, C).to(DEV), B, H, W, shift).cpu()
        assert torch.equal(back.view(B, H, W, C), t + t)            # gather then scatter is the identity on the crop
    m, H2, W2 = ops.swin_patch_merge(t.view(-1, C).to(DEV), B, H, W)
    xp = F.pad(t, (0, 0, 0, W % 2, 0, H % 2))
    ref = torch.cat([xp[:, 0::2, 0::2], xp[:, 1::2, 0::2], xp[:, 0::2, 1::2], xp[:, 1::2, 1::2]], -1)
    assert (H2, W2) == (6, 9) and torch.equal(m.cpu().view(B, H2, W2, 4 * C), ref)
    img = torch.randn(B, 10, 13, 4, generator=g)
    rows, Hp, Wp = ops.swin_patchify(img.to(DEV))
    ip = F.pad(img, (0, 0, 0, 3, 0, 2))
    ref = ip.view(B, 3, 4, 4, 4, 4).permute(0, 1, 3, 2, 4, 5).reshape(-1, 64)
    assert (Hp, Wp) == (3, 4) and torch.equal(rows.cpu(), ref)


def test_window_attention_vs_torch():
    from gomatching_amd import ops
    g = torch.Generator().manual_seed(1)
    heads, C, nW, B = 6, 192, 6, 2
    qkv = torch.randn(B * nW * 49, 3 * C, generator=g)
    bias = torch.randn(heads, 49, 49, generator=g)
    mask = swin_oracle.shift_mask(12, 17)
    for m in (None, mask):
        out = ops.swin_window_attention(qkv.to(DEV), bias.to(DEV), None if m is None else m.contiguous().to(DEV), nW,
                                        heads).cpu()
        t = qkv.view(B * nW, 49, 3, heads, 32).permute(2, 0, 3, 1, 4)
        a = (t[0] * 32 ** -0.5) @ t[1].transpose(-2, -1) + bias.unsqueeze(0)
        if m is not None:
            a = (a.view(B, nW, heads, 49, 49) + m.unsqueeze(1).unsqueeze(0)).view(-1, heads, 49, 49)
        ref = (a.softmax(-1) @ t[2]).transpose(1, 2).reshape(B * nW * 49, C)
        assert float((out - ref).abs().max()) <= 2e-5


@pytest.mark.parametrize("hw", [(96, 128), (90, 130)])
def test_swin_end_to_end_clip_vs_oracle(gemm_mode, hw):
    """The whole path with the Swin-T backbone (build_swin_backbone) on a 6-frame clip against the CPU oracle: identical
    ids and characters, points within 1e-3 px.  90x130 needs every internal padding of Swin (patch embed to 4, windows
    to 7, odd maps in the merges); the reference never pads the batch itself (gom_lstmatcher.py:169)."""
    from helpers import mini_cfg
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.synth import make_clip
    from oracle import gom_oracle as O
    cfg = mini_cfg("icdar15", device=DEV)
    cfg.MODEL.BACKBONE.NAME = "build_swin_backbone"
    sd = synth_state_dict(cfg, seed=7, cls_bias={"detection_transformer.ctrl_point_class.0.bias": 0.8,
                                                  "roi_heads.rescoring_head.bias": 0.8})
    clip = make_clip(6, hw[0], hw[1], clip_id=2)
    images = [torch.as_tensor(f.astype("float32").transpose(2, 0, 1)) for f in clip]
    ocfg = mini_cfg("icdar15")
    ocfg.MODEL.BACKBONE.NAME = "build_swin_backbone"
    with torch.no_grad():
        o_res, o_count = O.run_clip(sd, ocfg, images)
    model = GoMatching(cfg, sd, device=DEV, frames_per_step=3)
    tc = {k: 0.0 for k in ("pre_process", "backbone", "detector", "rescore", "tracker", "short_match", "long_match",
                           "post_process", "total_time")}
    insts, id_count = model.batch_inference([{"image": im, "height": hw[0], "width": hw[1]} for im in images], 0, 0, [], tc)
    insts = model._remove_short_track(insts)
    res = model.batch_postprocess(insts, [hw] * len(insts))
    dump = [(f, res[f]["instances"].track_ids.cpu().tolist(), o_res[f]["instances"]["track_ids"].tolist(),
             res[f]["instances"].scores.cpu().tolist(), o_res[f]["instances"]["scores"].tolist())
            for f in range(len(images))]
    assert int(id_count) == int(o_count), dump
    total = 0
    for f in range(len(images)):
        r, o = res[f]["instances"], o_res[f]["instances"]
        assert r.track_ids.cpu().tolist() == o["track_ids"].tolist(), dump
        assert r.recs.cpu().tolist() == o["recs"].tolist(), f
        assert float((r.bd.cpu() - o["bd"]).abs().max()) <= 1e-3 if len(r) else True
        assert float((r.scores.cpu() - o["scores"]).abs().max()) <= 1e-4 if len(r) else True
        total += len(r)
    assert total > 0


@pytest.mark.parametrize("graphs", [False, True])
@pytest.mark.parametrize("poison_value", [float("nan"), 1e30, -1e30])
@pytest.mark.parametrize("backbone", ["build_resnet_backbone", "build_swin_backbone"])
def test_no_kernel_reads_uninitialised_memory(backbone, poison_value, graphs, monkeypatch):
    """Every scratch / output buffer of the path comes from `torch.empty`.  Poison those allocations with NaN (floats) and
    a large sentinel (integers): the results must stay bit-identical to the unpoisoned run, at a frame size that leaves
    ragged tiles, padded windows and odd maps everywhere.  NaN catches arithmetic on stale memory, +-1e30 catches
    comparisons (fmaxf drops a NaN)."""
    from helpers import mini_cfg
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.synth import make_clip
    cfg = mini_cfg("icdar15", device=DEV)
    cfg.MODEL.BACKBONE.NAME = backbone
    sd = synth_state_dict(cfg, seed=7, cls_bias={"detection_transformer.ctrl_point_class.0.bias": 0.8,
                                                  "roi_heads.rescoring_head.bias": 0.8})
    hw = (90, 130)
    clip = make_clip(6, hw[0], hw[1], clip_id=2)
    images = [torch.as_tensor(f.astype("float32").transpose(2, 0, 1)) for f in clip]

    def run():
        model = GoMatching(cfg, sd, device=DEV, frames_per_step=3)
        model.use_graphs = graphs                                # (a fill inside a capture becomes a memset node)
        tc = {k: 0.0 for k in ("pre_process", "backbone", "detector", "rescore", "tracker", "short_match", "long_match",
                               "post_process", "total_time")}
        insts, idc = model.batch_inference([{"image": im, "height": hw[0], "width": hw[1]} for im in images], 0, 0, [], tc)
        return [(i.track_ids.cpu(), i.bd.cpu(), i.scores.cpu(), i.recs.cpu()) for i in insts], int(idc)

    clean, clean_count = run()
    real_empty, real_like = torch.empty, torch.empty_like

    def poison(x):
        if x.is_cuda:
            big = poison_value if x.dtype in (torch.float32, torch.float64) else max(-6e4, min(6e4, poison_value))
            # byte buffers are the split-K workspaces (fp32 partial sums behind a uint8 tensor): 0xFF bytes read back as NaN
            x.fill_(big if x.is_floating_point() else (1 << 30 if x.dtype in (torch.int32, torch.int64) else 255))
        return x

    monkeypatch.setattr(torch, "empty", lambda *a, **k: poison(real_empty(*a, **k)))
    monkeypatch.setattr(torch, "empty_like", lambda *a, **k: poison(real_like(*a, **k)))
    dirty, dirty_count = run()
    monkeypatch.undo()
    assert dirty_count == clean_count
    assert sum(len(c[0]) for c in clean) > 0
    for f, (c, d) in enumerate(zip(clean, dirty)):
        for a, b, name in zip(c, d, ("track_ids", "bd", "scores", "recs")):
            assert torch.equal(a, b), (f, name)
