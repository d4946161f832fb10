"""Seeded synthetic NV12 frames for tests (numpy; small sizes)."""
import numpy as np


def value_noise(rng, h, w, cell):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.random((gh, gw))
    ys, xs = np.arange(h) / cell, np.arange(w) / cell
    y0, x0 = ys.astype(int), xs.astype(int)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    a = g[y0][:, x0] * (1 - fx) + g[y0][:, x0 + 1] * fx
    b = g[y0 + 1][:, x0] * (1 - fx) + g[y0 + 1][:, x0 + 1] * fx
    return a * (1 - fy) + b * fy


def luma(seed, w, h, rects=40):
    """3-octave value noise (8 px base lattice) + bright/dark rectangles."""
    rng = np.random.default_rng(seed)
    img = sum(value_noise(rng, h, w, c) * a for c, a in ((8, 0.5), (16, 0.3), (32, 0.2)))
    img = 40 + 120 * img
    for _ in range(rects):
        rw, rh = rng.integers(max(4, w // 40), max(6, w // 12)), rng.integers(max(4, h // 40), max(6, h // 10))
        x, y = rng.integers(0, max(1, w - rw)), rng.integers(0, max(1, h - rh))
        img[y:y + rh, x:x + rw] = rng.choice([225.0, 30.0, 200.0])
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def nv12(seed, w, h, full_range=False):
    """Packed (h*3/2, w) NV12.  full_range=True uses every byte value (exercises saturation)."""
    rng = np.random.default_rng(seed + 7919)
    out = np.empty((h * 3 // 2, w), np.uint8)
    if full_range:
        out[:] = rng.integers(0, 256, out.shape, dtype=np.uint8)
        return out
    out[:h] = luma(seed, w, h)
    yy, xx = np.mgrid[0:h // 2, 0:w // 2]
    u = 128 + 60 * np.sin(xx / max(1, w // 2) * 3.1 + seed) * np.cos(yy / max(1, h // 2) * 2.3)
    v = 128 + 60 * np.cos(xx / max(1, w // 2) * 2.2) * np.sin(yy / max(1, h // 2) * 3.7 + seed)
    uv = out[h:].reshape(h // 2, w // 2, 2)
    uv[..., 0] = np.clip(np.rint(u), 0, 255)
    uv[..., 1] = np.clip(np.rint(v), 0, 255)
    return out


def shifted(img, dx, dy):
    """Sub-pixel translate a luma image by (dx, dy) with bilinear interpolation (float maths)."""
    h, w = img.shape
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    sx, sy = np.clip(xs - dx, 0, w - 1.001), np.clip(ys - dy, 0, h - 1.001)
    x0, y0 = sx.astype(int), sy.astype(int)
    fx, fy = sx - x0, sy - y0
    f = img.astype(np.float64)
    v = (f[y0, x0] * (1 - fx) + f[y0, x0 + 1] * fx) * (1 - fy) + (f[y0 + 1, x0] * (1 - fx) + f[y0 + 1, x0 + 1] * fx) * fy
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)
