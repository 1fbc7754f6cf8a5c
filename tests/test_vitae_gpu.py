"""ViTAEv2-S backbone on the GPU against the outputs of the reference's own ViTAEv2 module
(tests/golden/vitae_s.npz) under all three contraction back-ends; the glue kernels one by one against torch; the whole
path with this backbone against the CPU oracle."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import golden
from gomatching_amd.config import setup_cfg
from gomatching_amd.weights import synth_state_dict

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
    cfg.MODEL.BACKBONE.NAME = "build_vitaev2_backbone"
    return synth_state_dict(cfg, seed=3)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_vitae_matches_reference_module(tag, gemm_mode):
    from gomatching_amd.modeling.vitae import ViTAEv2S
    g = golden("vitae_s.npz")
    net = ViTAEv2S(_sd(), torch.device(DEV))
    x = torch.from_numpy(g["x_" + tag])
    x4 = torch.cat([x.permute(0, 2, 3, 1), x.new_zeros(x.shape[0], x.shape[2], x.shape[3], 1)], -1).contiguous().to(DEV)
    out = net.forward(x4)
    for k in ("stage3", "stage4", "stage5"):
        ref = torch.from_numpy(g["%s_%s" % (k, tag)])
        got = out[k].permute(0, 3, 1, 2).cpu()
        assert tuple(got.shape) == tuple(ref.shape)
        err = float((got - ref).abs().max())
        assert err <= 2e-5 * max(1.0, float(ref.abs().max())) * 5, (k, tag, err)      # 1e-4 at these magnitudes


def test_vitae_rejects_other_sizes():
    from gomatching_amd.modeling.vitae import ViTAEv2S
    net = ViTAEv2S(_sd(), torch.device(DEV))
    with pytest.raises(ValueError, match="multiples of 32"):
        net.forward(torch.zeros(1, 90, 130, 4, device=DEV))


def test_dilated_convolution_through_im2col():
    """PRM convolutions (kernel 7 stride 4 dilations 1-4; kernel 3 stride 2 dilations 1-3) = im2col + GEMM."""
    from gomatching_amd import ops
    g = torch.Generator().manual_seed(0)
    for (C, k, s, dils, H, W) in ((4, 7, 4, (1, 2, 3, 4), 32, 64), (64, 3, 2, (1, 2, 3), 16, 24)):
        x = torch.randn(2, C, H, W, generator=g)
        for d in dils:
            w = torch.randn(8, C, k, k, generator=g) * 0.1
            pad = math.ceil(((k - 1) * d + 1 - s) / 2)
            ref = F.conv2d(x, w, None, s, pad, d)
            K = k * k * C
            Kp = -(-K // 32) * 32
            cols, oh, ow = ops.im2col(x.permute(0, 2, 3, 1).contiguous().to(DEV), k, k, s, pad, d, Kp)
            assert (oh, ow) == tuple(ref.shape[-2:])
            assert Kp == K or float(cols[:, K:].abs().max()) == 0.0
            got = cols[:, :K].cpu() @ w.permute(0, 2, 3, 1).reshape(8, K).t()
            assert float((got.view(2, oh, ow, 8).permute(0, 3, 1, 2) - ref).abs().max()) <= 1e-4


def test_grouped_conv_bn_silu_vs_torch():
    from gomatching_amd import ops
    g = torch.Generator().manual_seed(1)
    for (cin, cout, groups, stride, H, W) in ((64, 64, 16, 2, 12, 16), (128, 512, 32, 1, 6, 8), (512, 128, 32, 1, 6, 8),
                                              (256, 256, 64, 1, 5, 3)):
        x = torch.randn(2, cin, H, W, generator=g)
        w = torch.randn(cout, cin // groups, 3, 3, generator=g) * 0.2
        scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
        conv = F.conv2d(x, w, None, stride, 1, 1, groups)
        ref = F.silu(conv * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
        xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        wd = w.permute(2, 3, 0, 1).contiguous().to(DEV)                       # [3,3,Cout,Cin/groups]
        got = ops.grouped_conv3x3(xd, wd, scale.to(DEV), shift.to(DEV), groups, stride=stride, silu=True)
        assert float((got.permute(0, 3, 1, 2).cpu() - ref).abs().max()) <= 1e-5
        R = torch.randn(ref.shape, generator=g)
        got = ops.grouped_conv3x3(xd, wd, None, shift.to(DEV), groups, stride=stride, silu=False,
                                  R=R.permute(0, 2, 3, 1).contiguous().to(DEV))
        ref2 = conv + shift.view(1, -1, 1, 1) + R
        assert float((got.permute(0, 3, 1, 2).cpu() - ref2).abs().max()) <= 1e-5
    v = torch.randn(1000, generator=g)
    assert float((ops.silu_(v.clone().to(DEV)).cpu() - F.silu(v)).abs().max()) <= 1e-6


@pytest.mark.parametrize("heads,hd", [(1, 64), (2, 64), (1, 128)])
def test_centred_window_attention_vs_torch(heads, hd):
    from gomatching_amd import ops
    g = torch.Generator().manual_seed(2)
    B, H, W, C = 2, 12, 16, heads * hd
    x = torch.randn(B, H, W, C, generator=g)
    td, lr = (7 - H % 7) % 7, (7 - W % 7) % 7
    top, left = td // 2, lr // 2
    xp = F.pad(x.permute(0, 3, 1, 2), (left, lr - left, top, td - top)).permute(0, 2, 3, 1)
    Hp, Wp = H + td, W + lr
    ref_win = xp.reshape(B, Hp // 7, 7, Wp // 7, 7, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, C)
    win = ops.vitae_window_gather(x.view(-1, C).to(DEV), B, H, W)
    assert torch.equal(win.cpu(), ref_win)
    qkv = torch.randn(win.shape[0], 3 * C, generator=g)
    q, k, v = qkv.view(-1, 49, 3, heads, hd).permute(2, 0, 3, 1, 4)
    ref = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(-1) @ v
    ref = ref.transpose(1, 2).reshape(-1, C)
    got = ops.vitae_window_attention(qkv.to(DEV), heads)
    assert float((got.cpu() - ref).abs().max()) <= 2e-5
    R1, R2 = torch.randn(B * H * W, C, generator=g), torch.randn(B * H * W, C, generator=g)
    crop = ref.view(B, Hp // 7, Wp // 7, 7, 7, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    crop = crop[:, top:top + H, left:left + W].reshape(-1, C)
    got = ops.vitae_window_crop(ref.to(DEV), B, H, W, R1=R1.to(DEV), R2=R2.to(DEV))
    assert float((got.cpu() - (crop + R1 + R2)).abs().max()) <= 1e-6
    assert torch.equal(ops.vitae_window_crop(ref.to(DEV), B, H, W).cpu(), crop)


def test_row_softmax_and_transpose():
    from gomatching_amd import ops
    g = torch.Generator().manual_seed(3)
    for rows, cols in ((5, 15), (48, 48), (3, 7168), (9, 1000)):
        ld = -(-cols // 4) * 4 + 4
        buf = torch.zeros(rows, ld)
        buf[:, :cols] = torch.randn(rows, cols, generator=g) * 3
        ref = (buf[:, :cols] * 0.125).softmax(-1)
        d = buf.to(DEV)
        ops.softmax_rows_scaled_(d, cols, 0.125)
        assert float((d[:, :cols].cpu() - ref).abs().max()) <= 1e-6
        assert float(d[:, cols:].abs().max()) == 0.0
    x = torch.randn(50, 3 * 64, generator=g).to(DEV)
    out = torch.zeros(64, 52, device=DEV)
    ops.transpose_into(x[:, 64:128], out)
    assert torch.equal(out[:, :50].cpu(), x[:, 64:128].t().cpu()) and float(out[:, 50:].abs().max()) == 0.0


@pytest.mark.parametrize("heads,hd", [(2, 64), (4, 64), (2, 128)])
@pytest.mark.parametrize("N", [15, 48, 64, 200, 1000])
def test_flash_attention_vs_torch(heads, hd, N):
    """softmax(q k^T / sqrt(hd)) v with the scores kept on the CU against torch fp64 -> fp32, ragged N (tail key tiles,
    tail query blocks), two images."""
    from gomatching_amd import ops
    g = torch.Generator().manual_seed(N + hd)
    B, C = 2, heads * hd
    qkv = torch.randn(B * N, 3 * C, generator=g)
    qkv[:, :C] *= 2.0                                            # scores with a real spread
    q, k, v = qkv.double().view(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    ref = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(-1) @ v
    ref = ref.transpose(1, 2).reshape(B * N, C).float()
    got = ops.flash_attention(qkv.to(DEV), B, N, heads).cpu()
    assert float((got - ref).abs().max()) <= 2e-5, float((got - ref).abs().max())
    ops.check_range_flag(DEV)


def test_flash_attention_operands_on_rounding_ties():
    """q, k, v values exactly half-way between two fp16 numbers (the case that cost a scalar split its low plane)."""
    import numpy as np
    from gomatching_amd import ops
    rng = np.random.default_rng(11)
    B, N, heads, hd = 1, 96, 2, 128
    C = heads * hd

    def ties(shape, e_lo, e_hi):
        e = rng.integers(e_lo, e_hi, size=shape).astype(np.float64)
        ulp = 2.0 ** (e - 10)
        return ((2.0 ** e + rng.integers(0, 1024, size=shape) * ulp + 0.5 * ulp) * rng.choice([-1.0, 1.0], size=shape)).astype(np.float32)

    qkv = torch.from_numpy(np.concatenate([ties((B * N, C), -4, -1), ties((B * N, C), -3, 1), ties((B * N, C), -6, 2)], 1))
    q, k, v = qkv.double().view(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    ref = (((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(-1) @ v).transpose(1, 2).reshape(B * N, C).float()
    got = ops.flash_attention(qkv.to(DEV), B, N, heads).cpu()
    assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max()), float((got - ref).abs().max())


def test_flash_and_gemm_pair_attention_agree():
    from gomatching_amd.modeling.vitae import ViTAEv2S
    g = golden("vitae_s.npz")
    net = ViTAEv2S(_sd(), torch.device(DEV))
    x = torch.from_numpy(g["x_a"])
    x4 = torch.cat([x.permute(0, 2, 3, 1), x.new_zeros(x.shape[0], x.shape[2], x.shape[3], 1)], -1).contiguous().to(DEV)
    a = net.forward(x4)
    net.flash = False
    b = net.forward(x4)
    for k in a:
        assert float((a[k] - b[k]).abs().max()) <= 1e-4, k


def test_vitae_end_to_end_clip_vs_oracle(gemm_mode):
    """The whole path with the ViTAEv2-S backbone on a 6-frame 96x128 clip against the CPU oracle: identical ids and
    characters, points within 1e-3 px; the ViTAE-specific post-process scale (gom_lstmatcher.py:82-96) included."""
    from helpers import mini_cfg
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.synth import make_clip
    from oracle import gom_oracle as O
    hw = (96, 128)
    cfgs = []
    for dev in (DEV, None):
        cfg = mini_cfg("icdar15", device=dev) if dev else mini_cfg("icdar15")
        cfg.MODEL.BACKBONE.NAME = "build_vitaev2_backbone"
        cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = 96, 200
        cfgs.append(cfg)
    cfg, ocfg = cfgs
    sd = synth_state_dict(cfg, seed=7, cls_bias={"detection_transformer.ctrl_point_class.0.bias": 0.8,
                                                  "roi_heads.rescoring_head.bias": 0.8})
    clip = make_clip(6, hw[0], hw[1], clip_id=2)
    images = [torch.as_tensor(f.astype("float32").transpose(2, 0, 1)) for f in clip]
    orig = (72, 96)                                              # the frames "were" 72x96 before the harness resize
    with torch.no_grad():
        o_res, o_count = O.run_clip(sd, ocfg, images, orig_hw=orig)
    model = GoMatching(cfg, sd, device=DEV, frames_per_step=3)
    tc = {k: 0.0 for k in ("pre_process", "backbone", "detector", "rescore", "tracker", "short_match", "long_match",
                           "post_process", "total_time")}
    insts, id_count = model.batch_inference([{"image": im, "height": orig[0], "width": orig[1]} for im in images], 0, 0, [],
                                            tc)
    insts = model._remove_short_track(insts)
    res = model.batch_postprocess(insts, [orig] * len(insts))
    assert int(id_count) == int(o_count)
    total = 0
    for f in range(len(images)):
        r, o = res[f]["instances"], o_res[f]["instances"]
        assert r.track_ids.cpu().tolist() == o["track_ids"].tolist(), f
        assert r.recs.cpu().tolist() == o["recs"].tolist(), f
        if len(r):
            assert float((r.bd.cpu() - o["bd"]).abs().max()) <= 1e-3
            assert float((r.ctrl_points.cpu() - o["ctrl_points"]).abs().max()) <= 1e-3
            assert float((r.scores.cpu() - o["scores"]).abs().max()) <= 1e-4
        total += len(r)
    assert total > 0
