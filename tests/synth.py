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


# ---------------------------------------------------------------------------------------------
# synthetic shaky clips with known camera rotations (fisheye forward model)
# ---------------------------------------------------------------------------------------------
def sphere_texture(seed, tw=2048, th=1024):
    """Equirectangular luma texture: value noise + rectangles (trackable corners)."""
    return luma(seed, tw, th, rects=260).astype(np.float32)


def fisheye_rays(K, w, h):
    """Unit rays of every pixel of an equidistant fisheye camera (theta = |(x-c)/f|)."""
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    px, py = (xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1]
    th = np.hypot(px, py)
    s = np.where(th > 1e-12, np.sin(th) / np.maximum(th, 1e-12), 1.0)
    return np.stack([px * s, py * s, np.cos(th)], axis=-1)


def pinhole_rays(K, w, h):
    """Unit rays of every pixel of a pinhole camera."""
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    d = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones((h, w))], axis=-1)
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


def fisheye_project(K, rays):
    """Inverse of fisheye_rays for arbitrary rays (n,3) -> pixels (n,2)."""
    x, y, z = rays[:, 0], rays[:, 1], rays[:, 2]
    r = np.hypot(x, y)
    th = np.arctan2(r, z)
    s = np.where(r > 1e-12, th / np.maximum(r, 1e-12), 1.0)
    return np.stack([K[0, 2] + K[0, 0] * x * s, K[1, 2] + K[1, 1] * y * s], axis=-1)


def render_frame(tex, rays, R):
    """Luma of the frame seen by a camera with orientation R (d_cam = R d_world)."""
    d = rays @ R  # rows: R^T d_cam
    lon = np.arctan2(d[..., 0], d[..., 2])
    lat = np.arcsin(np.clip(d[..., 1], -1, 1))
    th, tw = tex.shape
    u = (lon / (2 * np.pi) + 0.5) * (tw - 1)
    v = (lat / np.pi + 0.5) * (th - 1)
    u0, v0 = np.clip(u.astype(int), 0, tw - 2), np.clip(v.astype(int), 0, th - 2)
    fu, fv = u - u0, v - v0
    val = (tex[v0, u0] * (1 - fu) + tex[v0, u0 + 1] * fu) * (1 - fv) + (tex[v0 + 1, u0] * (1 - fu) + tex[v0 + 1, u0 + 1] * fu) * fv
    return np.clip(np.rint(val), 0, 255).astype(np.uint8)


def shaky_clip(seed, K, w, h, n, sigma=0.004, projection="fish"):
    """n packed NV12 frames + the camera orientations R_k (random walk, sigma rad/frame/axis)."""
    import oracle
    rng = np.random.default_rng(seed)
    tex = sphere_texture(seed)
    rays = fisheye_rays(K, w, h) if projection == "fish" else pinhole_rays(K, w, h)
    R = np.eye(3)
    frames, rots = [], []
    for k in range(n):
        if k:
            R = oracle.rodrigues(rng.normal(0, sigma, 3)) @ R
        f = np.empty((h * 3 // 2, w), np.uint8)
        f[:h] = render_frame(tex, rays, R)
        f[h:] = 128
        f[h:, ::2] = (100 + 40 * np.sin(np.arange(w // 2) / 17.0 + k)).astype(np.uint8)[None, :]
        frames.append(f)
        rots.append(R.copy())
    return frames, rots


def edge_leaving_pair(seed, w=160, h=120, n=60):
    """A smooth random texture and the same texture shifted by 6..22 px towards the right (even seeds) or the bottom
    (odd seeds) edge, plus n feature points in the 14-px band along that edge: inputs on which features are tracked
    out of the image -- some by the last Gauss-Newton step only, which is where OpenCV's test of the final position
    (LKTrackerInvoker behind its loop) decides.  -> (prev u8, next u8, pts float32 (n, 2))."""
    from scipy.ndimage import gaussian_filter, shift as nd_shift
    rng = np.random.default_rng(1000 + seed)
    base = rng.integers(0, 256, (h // 4 + 13, w // 4 + 13)).astype(np.float32)
    img = gaussian_filter(np.kron(base, np.ones((4, 4), np.float32))[:h + 40, :w + 40], 1.5)
    along, across = rng.uniform(6, 22), rng.uniform(-3, 3)
    sx, sy = (along, across) if seed % 2 == 0 else (across, along)
    prev = np.clip(img[:h, :w], 0, 255).astype(np.uint8)
    nxt = np.clip(nd_shift(img, (sy, sx), order=1, mode="nearest")[:h, :w], 0, 255).astype(np.uint8)
    if seed % 2 == 0:
        pts = np.stack([rng.uniform(w - 14, w - 1, n), rng.uniform(5, h - 5, n)], 1)
    else:
        pts = np.stack([rng.uniform(5, w - 5, n), rng.uniform(h - 14, h - 1, n)], 1)
    return prev, nxt, pts.astype(np.float32)
