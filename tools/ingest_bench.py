"""Time the frame-ingest kernel (u8 HWC frames -> normalised fp32 NHWC4 backbone input) at BASELINE's frame size and
the host path it replaces (PIL resize + astype + H2D of fp32).  Prints one JSON line."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from gomatching_amd import ops                                   # noqa: E402
from gomatching_amd.predictor import resize_shortest_edge, resized_shape   # noqa: E402


def main(frames=8, src=(720, 1280), mn=1000, mx=2000, iters=50):
    g = np.random.default_rng(0)
    clip = g.integers(0, 256, size=(frames,) + src + (3,), dtype=np.uint8)
    oh, ow = resized_shape(src[0], src[1], mn, mx)
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    dev = torch.device("cuda:0")
    pinned = torch.as_tensor(clip).pin_memory()
    u8 = pinned.to(dev)
    for _ in range(3):
        ops.ingest(u8, oh, ow, mean, std, True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        ops.ingest(u8, oh, ow, mean, std, True)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / iters
    alg = frames * (src[0] * src[1] * 3 + oh * ow * 16)
    t0 = time.time()
    for _ in range(5):
        u8 = pinned.to(dev, non_blocking=True)
        ops.ingest(u8, oh, ow, mean, std, True)
        torch.cuda.synchronize()
    dev_path = (time.time() - t0) / 5
    t0 = time.time()
    host = [torch.as_tensor(resize_shortest_edge(np.ascontiguousarray(f[:, :, ::-1]), mn, mx).astype("float32")
                            .transpose(2, 0, 1)) for f in clip]
    x = torch.stack([h.to(dev) for h in host]).contiguous()
    ops.preprocess(x, mean, std)
    torch.cuda.synchronize()
    host_path = time.time() - t0
    print(json.dumps({"kernel": "resample_kernel<f32>", "frames": frames, "src": src, "dst": (oh, ow),
                      "us_per_launch": round(us, 1), "algorithmic_MB": round(alg / 1e6, 1),
                      "GBps": round(alg / us / 1e3, 1), "frac_of_8TBps": round(alg / us / 1e3 / 8000, 3),
                      "device_path_ms_incl_h2d_u8": round(dev_path * 1e3, 2),
                      "host_path_ms_pil_plus_h2d_f32": round(host_path * 1e3, 1)}))


if __name__ == "__main__":
    main()
