"""Synthetic clips for parity tests and the benchmark (SURVEY.md §8-d "Synthetic inputs").

Frames are uint8 RGB: seed-indexed smooth noise plus K moving high-contrast rectangles so that
detections persist across frames.  Generator: numpy Philox keyed by 0x60A7 + clip_id.
"""
import numpy as np


def make_clip(num_frames, height, width, clip_id=0, num_rects=6):
    rng = np.random.Generator(np.random.Philox(key=0x60A7 + clip_id))
    # smooth background: low-res noise upsampled by pixel replication + a gradient
    gh, gw = max(2, height // 32), max(2, width // 32)
    low = rng.uniform(60, 160, size=(gh, gw, 3)).astype(np.float32)
    yy = (np.arange(height) * gh // height).clip(0, gh - 1)
    xx = (np.arange(width) * gw // width).clip(0, gw - 1)
    base = low[yy][:, xx]
    base += np.linspace(0, 30, width, dtype=np.float32)[None, :, None]
    rects = []
    for _ in range(num_rects):
        w = int(rng.integers(width // 10, width // 4))
        h = int(rng.integers(height // 14, height // 6))
        x = float(rng.uniform(0, width - w))
        y = float(rng.uniform(0, height - h))
        vx, vy = rng.uniform(-0.01, 0.01, size=2) * (width, height)
        fg = rng.uniform(200, 255, size=3).astype(np.float32)
        bg = rng.uniform(0, 40, size=3).astype(np.float32)
        rects.append((x, y, w, h, vx, vy, fg, bg))
    frames = []
    for t in range(num_frames):
        img = base + rng.uniform(-3, 3, size=base.shape).astype(np.float32)
        for (x, y, w, h, vx, vy, fg, bg) in rects:
            x0 = int(np.clip(x + vx * t, 0, width - w))
            y0 = int(np.clip(y + vy * t, 0, height - h))
            img[y0:y0 + h, x0:x0 + w] = bg
            # "glyph" stripes
            for s in range(x0 + 2, x0 + w - 2, max(4, w // 8)):
                img[y0 + h // 5: y0 + h - h // 5, s:s + max(2, w // 16)] = fg
        frames.append(np.clip(img, 0, 255).astype(np.uint8))
    return frames
