"""Diagnostic (CPU): how far is the fp32 CPU oracle from its own float64 evaluation on one full-size frame?  That distance is the
floor of any fp32-vs-fp32 parity tolerance.  python tools/diag/oracle_f64.py <builtin> <src H> <src W> <clip id>
Measured in the build container: icdar15 720x1280 -> bd 4.4e-7 normalised (0.6e-3 px); pp_dstext 1080x1920 -> 1280x2276:
bd 4.2e-7 (0.95e-3 px), ctrl 3.8e-7, point logits 8.5e-6, query features 2.5e-5."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gomatching_amd.config import setup_cfg
from gomatching_amd.weights import synth_state_dict
from gomatching_amd.synth import make_clip
from gomatching_amd.predictor import GoMBatchPredictor
from oracle import gom_oracle as O
builtin, H, W, cid = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = setup_cfg(builtin=builtin); cfg.MODEL.DEVICE = 'cpu'
frames = make_clip(1, H, W, clip_id=cid, num_rects=14)
x, hw = GoMBatchPredictor(cfg, None).prepare([frames[0][:, :, ::-1]])
im = x[0]['image']
print('net size', tuple(im.shape))
sd = synth_state_dict(cfg, seed=2)
torch.set_num_threads(8)
res = {}
for name, dt in (('f32', torch.float32), ('f64', torch.float64)):
    sdd = {k: (torch.as_tensor(v).to(dt) if torch.as_tensor(v).is_floating_point() else torch.as_tensor(v)) for k, v in sd.items()}
    taps = {}
    t0 = time.time()
    with torch.no_grad():
        O.detect_frames(sdd, cfg, [im.to(dt)], taps=taps)
    print(name, 'took %.1f s' % (time.time() - t0), flush=True)
    res[name] = {k: v.double() for k, v in taps.items() if k.startswith('out_') or k in ('res5',)}
for k in res['f32']:
    a, b = res['f32'][k], res['f64'][k]
    print(k, tuple(a.shape), 'max|f32-f64| = %.3e   (max|f64| %.3e)' % (float((a - b).abs().max()), float(b.abs().max())))
bd = (res['f32']['out_pred_bd_points'] - res['f64']['out_pred_bd_points']).abs()
print('bd err px (x scaled by W):', float(bd[..., 0::2].max()) * im.shape[-1], float(bd[..., 1::2].max()) * im.shape[-2])
