"""Deterministic synthetic frames: mid-grey field + Gaussian blobs + noise,
quantised to u8 and replicated to BGRA8.  Shared by tests and bench.py."""
import numpy as np

BASE_SEED = 20250321


def blob_frame(width, height, frame_index=0, n_blobs=None, gray=False):
    rng = np.random.default_rng(BASE_SEED + frame_index)
    P = width * height
    if n_blobs is None:
        n_blobs = max(8, int(round(4000 * P / 2.0736e6)))
    img = np.full((height, width), 0.5, np.float32)
    cx = rng.uniform(0, width, n_blobs)
    cy = rng.uniform(0, height, n_blobs)
    sg = np.exp(rng.uniform(np.log(1.5), np.log(24.0), n_blobs))
    am = rng.uniform(0.08, 0.35, n_blobs) * rng.choice([-1.0, 1.0], n_blobs)
    for i in range(n_blobs):
        r = int(np.ceil(4 * sg[i]))
        x0, x1 = max(0, int(cx[i]) - r), min(width, int(cx[i]) + r + 1)
        y0, y1 = max(0, int(cy[i]) - r), min(height, int(cy[i]) + r + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        xs = np.arange(x0, x1, dtype=np.float32) - np.float32(cx[i])
        ys = np.arange(y0, y1, dtype=np.float32) - np.float32(cy[i])
        gx = np.exp(-0.5 * (xs / sg[i]) ** 2).astype(np.float32)
        gy = np.exp(-0.5 * (ys / sg[i]) ** 2).astype(np.float32)
        img[y0:y1, x0:x1] += np.float32(am[i]) * gy[:, None] * gx[None, :]
    img += rng.normal(0, 0.01, img.shape).astype(np.float32)
    u8 = np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8)
    if gray:
        return u8
    bgra = np.empty((height, width, 4), np.uint8)
    bgra[..., 0] = u8
    bgra[..., 1] = u8
    bgra[..., 2] = u8
    bgra[..., 3] = 255
    return bgra
