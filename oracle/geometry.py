"""Host-geometry part of the CPU oracle (numpy, fp64) -- TEST INFRASTRUCTURE ONLY.

File:line citations are into /root/reference/opencv/.
"""
import math
import sys

import numpy as np

# CameraPreset, FrameSourceWarp.hpp:14-21
GOPRO_H4B_WIDE43_PUBLISHED = 0
GOPRO_H4B_WIDE43_MEASURED = 1
GOPRO_H4B_WIDE43_MEASURED_STABILISATION = 2
GOPRO_H4B_WIDE169_PUBLISHED = 3
GOPRO_H4B_WIDE169_MEASURED = 4
GOPRO_H4B_WIDE169_MEASURED_STABILISATION = 5

# FrameSourceWarp.cpp:22-25 -- declared `const int`, so the fractional part is dropped
_FOV_H_43W = int(122.6)
_FOV_V_43W = int(94.4)
_FOV_H_169W = int(118.2)
_FOV_V_169W = int(69.5)


def get_preset_camera(preset, width, height):
    """get_preset_camera, FrameSourceWarp.cpp:27-86.  Returns the 3x3 fp64 camera matrix."""
    K = np.eye(3)
    K[0, 2] = (width - 1.0) / 2
    K[1, 2] = (height - 1.0) / 2
    if preset == GOPRO_H4B_WIDE43_PUBLISHED:
        K[0, 0] = width / (_FOV_H_43W * math.pi / 180)
        K[1, 1] = height / (_FOV_V_43W * math.pi / 180)
    elif preset == GOPRO_H4B_WIDE169_PUBLISHED:
        K[0, 0] = width / (_FOV_H_169W * math.pi / 180)
        K[1, 1] = height / (_FOV_V_169W * math.pi / 180)
    elif preset == GOPRO_H4B_WIDE43_MEASURED:
        K[0, 2] = 967.37 * width / 1920
        K[1, 2] = 711.07 * height / 1440
        K[0, 0] = 942.96 * height / 1440
        K[1, 1] = 942.53 * height / 1440
    elif preset == GOPRO_H4B_WIDE43_MEASURED_STABILISATION:
        K[0, 2] = 965.90 * width / 1920
        K[1, 2] = 712.94 * height / 1440
        K[0, 0] = 1045.58 * height / 1440
        K[1, 1] = 1045.64 * height / 1440
    elif preset == GOPRO_H4B_WIDE169_MEASURED:
        K[0, 2] = 1361.80 * width / 2704
        K[1, 2] = 745.19 * height / 1520
        K[0, 0] = 1392.49 * height / 1520
        K[1, 1] = 1383.47 * height / 1520
    elif preset == GOPRO_H4B_WIDE169_MEASURED_STABILISATION:
        K[0, 2] = 1357.49 * width / 2704
        K[1, 2] = 736.74 * height / 1520
        K[0, 0] = 1626.67 * height / 1520
        K[1, 1] = 1619.46 * height / 1520
    else:
        raise ValueError("unknown preset")
    return K


PROJ_RECT, PROJ_FISH = 0, 1


def lens_camera(projection, dfov_deg, width, height, cx=None, cy=None):
    """Camera matrix of a libdewobble-style lens (render.ts:611-617,669-683): projection + diagonal field of
    view; principal point defaults to (width/2, height/2) (render.ts:682-683).  Defined by this project
    (libdewobble is not in the reference tree)."""
    half_diag = 0.5 * math.hypot(width, height)
    half_fov = 0.5 * math.radians(dfov_deg)
    f = half_diag / math.tan(half_fov) if projection == PROJ_RECT else half_diag / half_fov
    K = np.eye(3)
    K[0, 0] = K[1, 1] = f
    K[0, 2] = width / 2 if cx is None else cx
    K[1, 2] = height / 2 if cy is None else cy
    return K


def map_mode(in_projection, out_projection):
    """vstab_map_mode of a projection pair: 1 fish->rect, 2 fish->fish, 3 rect->rect, 4 rect->fish."""
    if in_projection == PROJ_FISH:
        return 2 if out_projection == PROJ_FISH else 1
    return 4 if out_projection == PROJ_FISH else 3


def fisheye_undistort_points(pts, K, R=None, P=None):
    """cv::fisheye::undistortPoints with D = 0 (calls at FrameSourceWarp.cpp:93,322,333).

    Third-party arithmetic (OpenCV 4.5 calib3d/fisheye.cpp), fp64.
    """
    pts = np.asarray(pts, np.float64).reshape(-1, 2)
    RR = np.eye(3) if R is None else np.asarray(R, np.float64)
    if P is not None:
        RR = np.asarray(P, np.float64)[:3, :3] @ RR
    out = np.empty_like(pts)
    for i, (x, y) in enumerate(pts):
        pw = np.array([(x - K[0, 2]) / K[0, 0], (y - K[1, 2]) / K[1, 1]])
        theta_d = math.sqrt(pw[0] * pw[0] + pw[1] * pw[1])
        theta_d = min(max(-math.pi / 2, theta_d), math.pi / 2)
        scale = 0.0
        if abs(theta_d) > 1e-8:
            # Newton iteration on theta*(1 + k.theta^2..) = theta_d is the identity for k = 0
            scale = math.tan(theta_d) / theta_d
        pu = pw * scale
        pr = RR @ np.array([pu[0], pu[1], 1.0])
        out[i] = (pr[0] / pr[2], pr[1] / pr[2])
    return out


def _cv_round(v):
    """cvRound / saturate_cast<int>(double): round half to even."""
    return int(np.rint(v))


def get_output_camera(K_in, width, height, scale=1.0, crop_borders=False, zoom=1.0):
    """get_output_camera, FrameSourceWarp.cpp:88-165.  Returns (K_out, (out_w, out_h))."""
    probes = [
        (0, 0), (0, height - 1), (width - 1, 0), (width - 1, height - 1),
        (K_in[0, 2], 0), (width - 1, K_in[1, 2]), (K_in[0, 2], height - 1), (0, K_in[1, 2]),
    ]
    ext = fisheye_undistort_points(probes, K_in)
    start = 4 if crop_borders else 0
    max_x, min_x = ext[start:, 0].max(), ext[start:, 0].min()
    max_y, min_y = ext[start:, 1].max(), ext[start:, 1].min()
    idx, idy = _cv_round(width - 1), _cv_round(height - 1)          # :142 cv::Point(int)
    in_len = math.sqrt(1.0 * idx * idx + idy * idy)
    d = ext[3] - ext[0]
    odx, ody = _cv_round(d[0]), _cv_round(d[1])                      # :146 cv::Point(int)
    out_len = math.sqrt(1.0 * odx * odx + ody * ody)
    scale = scale * (in_len / out_len)
    K = np.eye(3)
    K[0, 0] = scale
    K[1, 1] = scale
    K[0, 2] = scale * -min_x / zoom
    K[1, 2] = scale * -min_y / zoom
    size = (int(scale * (max_x - min_x) / zoom), int(scale * (max_y - min_y) / zoom))  # :163 trunc
    return K, size


def map_params(K_in, K_out, R):
    """The 17 cl_float kernel arguments, FrameSourceWarp.cpp:283-299 (double -> float casts)."""
    R = np.asarray(R, np.float64)
    return np.array([K_in[0, 2], K_in[1, 2], K_in[0, 0], K_in[1, 1],
                     K_out[0, 2], K_out[1, 2], K_out[0, 0], K_out[1, 1],
                     *R.reshape(-1)], np.float64).astype(np.float32)


def rodrigues(rvec):
    """Rotation vector -> matrix (cv::Rodrigues as used at FrameSourceWarp.cpp:373)."""
    rvec = np.asarray(rvec, np.float64).reshape(3)
    th = np.linalg.norm(rvec)
    if th < 1e-300:
        return np.eye(3)
    k = rvec / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return math.cos(th) * np.eye(3) + (1 - math.cos(th)) * np.outer(k, k) + math.sin(th) * Kx


def gyro_integrate(samples, rate_scale, t_prev_first_row, t_first_row, t_last_row):
    """What the reference's stubbed gyro path has to compute (gpmf.cpp:5-11 GyroFrame {start_ts, end_ts, roll, pitch, yaw};
    AvFrameSourceFileVaapi.cpp:121-123): per span, the ordered product of exponential maps of rate x overlap, later samples
    on the left (the accumulation order of FrameSourceWarp.cpp:441).  -> (R_delta over [t_prev_first_row, t_first_row],
    R_readout over [t_first_row, t_last_row]).  Independent numpy restatement of vstab_gyro_integrate."""
    s = np.asarray(samples, np.float64).reshape(-1, 5)

    def span(a, b):
        R = np.eye(3)
        for start, end, roll, pitch, yaw in s:
            lo, hi = max(start, a), min(end, b)
            if hi > lo:
                R = rodrigues(np.array([pitch, yaw, roll]) * ((hi - lo) * rate_scale)) @ R
        return R
    return span(t_prev_first_row, t_first_row), span(t_first_row, t_last_row)


def rotation_angle(R):
    return math.acos(max(-1.0, min(1.0, (np.trace(R) - 1) / 2)))


# ---------------------------------------------------------------------------------------------
# gram_sg::RotationFilter(SavitzkyGolayFilterConfig(m, t=0, n=2, s=0)); calls at
# FrameSourceWarp.cpp:212,444,459,471.  Third-party arithmetic (arntanguy/gram_savitzky_golay,
# unversioned, meson.build:37): Gorry's Gram-polynomial recursion.
# ---------------------------------------------------------------------------------------------
def _gram_poly(i, m, k, s):
    if k > 0:
        return ((4.0 * k - 2.0) / (k * (2.0 * m - k + 1.0)) *
                (i * _gram_poly(i, m, k - 1, s) + s * _gram_poly(i, m, k - 1, s - 1)) -
                ((k - 1.0) * (2.0 * m + k)) / (k * (2.0 * m - k + 1.0)) * _gram_poly(i, m, k - 2, s))
    return 1.0 if (k == 0 and s == 0) else 0.0


def _gen_fact(a, b):
    g = 1.0
    for j in range(a - b + 1, a + 1):
        g *= j
    return g


def sg_weights(m, t=0, n=2, s=0):
    w = np.zeros(2 * m + 1)
    for i in range(-m, m + 1):
        acc = 0.0
        for k in range(n + 1):
            acc += ((2 * k + 1) * (_gen_fact(2 * m, k) / _gen_fact(2 * m + k + 1, k + 1)) *
                    _gram_poly(i, m, k, 0) * _gram_poly(t, m, k, s))
        w[i + m] = acc
    return w


class RotationFilter:
    """Ring of 2m+1 3x3 matrices initialised to ZERO; filter() = polar factor U.V^T of the
    weighted sum (no determinant fix)."""

    def __init__(self, m):
        self.m = m
        self.w = sg_weights(m)
        self.buf = [np.zeros((3, 3)) for _ in range(2 * m + 1)]

    def add(self, R):
        self.buf.pop(0)
        self.buf.append(np.array(R, np.float64))

    def filter(self):
        M = np.zeros((3, 3))
        for wi, Ri in zip(self.w, self.buf):
            M += wi * Ri
        U, _, Vt = np.linalg.svd(M)
        return U @ Vt


def rodrigues_inv(R):
    """Rotation matrix -> rotation vector (cv::Rodrigues, matrix input), angles below pi."""
    R = np.asarray(R, np.float64)
    c = min(max((np.trace(R) - 1) / 2, -1.0), 1.0)
    th = math.acos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-12:
        return v / 2
    return v * (th / (2 * math.sin(th)))


class KalmanRotationFilter:
    """The optional Kalman smoother: one cv::KalmanFilter(2, 1, 0) per rotation-vector axis with the
    constants of init_filter (FrameSourceWarp.cpp:167-175 == kalman/kalman.cpp:43-48): F = [[1,1],[0,1]], H = [1 0],
    Q = 1e-5 I, R = 1e-1, P0 = I, x0 = 0; every frame predict() then correct(measured angle), output = corrected
    state.  Written in the general matrix form of cv::KalmanFilter::predict / correct.  The reference never calls
    init_filter, so this mode has no reference output."""

    def __init__(self):
        self.F = np.array([[1.0, 1.0], [0.0, 1.0]])
        self.H = np.array([[1.0, 0.0]])
        self.Q = 1e-5 * np.eye(2)
        self.Rn = np.array([[1e-1]])
        self.x = [np.zeros((2, 1)) for _ in range(3)]
        self.P = [np.eye(2) for _ in range(3)]

    def update(self, measured):
        rv = rodrigues_inv(measured)
        out = np.zeros(3)
        for a in range(3):
            x = self.F @ self.x[a]                                   # predict
            P = self.F @ self.P[a] @ self.F.T + self.Q
            S = self.H @ P @ self.H.T + self.Rn                      # correct
            K = P @ self.H.T @ np.linalg.inv(S)
            x = x + K @ (np.array([[rv[a]]]) - self.H @ x)
            P = P - K @ self.H @ P
            self.x[a], self.P[a] = x, P
            out[a] = x[0, 0]
        return rodrigues(out)


# ---------------------------------------------------------------------------------------------
# guess_camera_rotation, FrameSourceWarp.cpp:316-375, restated independently of the product (numpy, fp64).
#
# What the reference does: undistort the current points into output-camera pixels (:322-330) and the
# previous points into normalised coordinates (:333-338), give every previous point a random depth
# s = rand() / RAND_MAX -> object point (x s, y s, s) (:345-350), solvePnPRansac(100 iterations, 8 px, 0.99)
# (:354-366), Rodrigues of the result (:373); a solver exception becomes (identity, 0 inliers) (:367-371).
#
# What is restated here rather than taken from OpenCV 4.5 (not in the image):
#   * rand() is un-seeded libc; this project seeds PCG32 (O'Neill) instead -- one stream per clip, one
#     uniform draw per point, then the RANSAC draws.
#   * cv::solvePnPRansac = RANSACPointSetRegistrator (calib3d/ptsetreg.cpp): subsets of 5 distinct points,
#     model from the minimal solver, inliers by squared reprojection error (float) <= threshold^2, best
#     model = most inliers (> max(best, 4)), iteration bound updated by RANSACUpdateNumIters; then a refit
#     on the inliers (SOLVEPNP_ITERATIVE: Levenberg-Marquardt on the reprojection error).
#   * DEVIATION: OpenCV's minimal solver for 5 points is EPnP; here (and in the product) the minimal solver
#     is the same Levenberg-Marquardt iteration started from the identity pose (10 steps), which is what
#     the data supports (inter-frame rotations of a few degrees).  tests/test_motion_cpu.py measures how far
#     that carries (inter-frame rotations up to 8 degrees, 50 % outliers).
# ---------------------------------------------------------------------------------------------
class Pcg32:
    """PCG32 XSH-RR (pcg-random.org minimal C implementation), seed/sequence initialisation included."""
    M64 = (1 << 64) - 1

    def __init__(self, seed=42, seq=54):
        self.state, self.inc = 0, ((seq << 1) | 1) & self.M64
        self.next()
        self.state = (self.state + seed) & self.M64
        self.next()

    def next(self):
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & self.M64
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xFFFFFFFF

    def uniform(self):  # [0, 1], like rand() * 1. / RAND_MAX
        return self.next() * (1.0 / 4294967295.0)

    def below(self, n):
        return (self.next() * n) >> 32


def _pnp_project(R, t, X, f, cx, cy):
    Y = X @ R.T + t
    return np.stack([f * Y[:, 0] / Y[:, 2] + cx, f * Y[:, 1] / Y[:, 2] + cy], axis=1)


def _pnp_cost(R, t, X, u, f, cx, cy):
    with np.errstate(all="ignore"):
        e = _pnp_project(R, t, X, f, cx, cy) - u
        return float(np.sum(e * e))


def _solve_pnp_lm(X, u, f, cx, cy, R, t, max_iter):
    """Levenberg-Marquardt on the reprojection error over a left rotation increment w and a translation
    increment: u = f Yx / Yz + cx, v = f Yy / Yz + cy, Y = R X + t, dY/dw = -[R X]x, dY/dt = I.
    Damping: (J^T J + lambda diag(J^T J)) d = -J^T r, lambda x 0.1 on success, x 10 on failure (at most 8
    tries per step); stops when a step gains less than 1e-9 of the cost."""
    lam = 1e-3
    cost = _pnp_cost(R, t, X, u, f, cx, cy)
    if not math.isfinite(cost):
        return None
    for _ in range(max_iter):
        RX = X @ R.T
        Y = RX + t
        iz = 1.0 / Y[:, 2]
        fz, xz, yz = f * iz, Y[:, 0] * iz, Y[:, 1] * iz
        eu, ev = fz * Y[:, 0] + cx - u[:, 0], fz * Y[:, 1] + cy - u[:, 1]
        x, y, z = RX[:, 0], RX[:, 1], RX[:, 2]
        zero = np.zeros_like(x)
        Ju = np.stack([-fz * xz * y, fz * (z + xz * x), -fz * y, fz, zero, -fz * xz], axis=1)
        Jv = np.stack([-fz * (z + yz * y), fz * yz * x, fz * x, zero, fz, -fz * yz], axis=1)
        H = Ju.T @ Ju + Jv.T @ Jv
        g = Ju.T @ eu + Jv.T @ ev
        improved = False
        for _try in range(8):
            A = H + lam * np.diag(np.diag(H) + 1e-12)
            try:
                d = np.linalg.solve(A, -g)
            except np.linalg.LinAlgError:
                lam *= 10
                continue
            Rc, tc = rodrigues(d[:3]) @ R, t + d[3:]
            c2 = _pnp_cost(Rc, tc, X, u, f, cx, cy)
            if math.isfinite(c2) and c2 < cost:
                rel = (cost - c2) / max(cost, 1e-300)
                R, t, cost, lam, improved = Rc, tc, c2, max(lam * 0.1, 1e-12), True
                if rel < 1e-9 or cost < 1e-20:
                    return R, t
                break
            lam *= 10
        if not improved:
            break
    return R, t


def _ransac_update_iters(p, ep, model_points, max_iters):
    """cv::RANSACUpdateNumIters (calib3d/ptsetreg.cpp)."""
    p, ep = min(max(p, 0.0), 1.0), min(max(ep, 0.0), 1.0)
    num, denom = max(1.0 - p, sys.float_info.min), 1.0 - (1.0 - ep) ** model_points
    if denom < sys.float_info.min:
        return 0
    num, denom = math.log(num), math.log(denom)
    return max_iters if denom >= 0 or -num >= max_iters * (-denom) else int(np.rint(num / denom))


def estimate_rotation(prev_pts, cur_pts, K_in, K_out, rng, in_fish=True):
    """-> (R, number of RANSAC inliers).  prev_pts / cur_pts: (n, 2) float32 input-image pixels;
    rng: Pcg32 (the clip's stream; it is advanced)."""
    prev_pts = np.asarray(prev_pts, np.float32).reshape(-1, 2)
    cur_pts = np.asarray(cur_pts, np.float32).reshape(-1, 2)
    n = len(prev_pts)
    if n < 5:
        return np.eye(3), 0
    if in_fish:
        und_cur = fisheye_undistort_points(cur_pts, K_in, None, K_out)
        und_prev = fisheye_undistort_points(prev_pts, K_in)
    else:  # pinhole input lens (the libdewobble in_p=rect surface): no tan(theta) / theta factor
        nc = (cur_pts.astype(np.float64) - [K_in[0, 2], K_in[1, 2]]) / [K_in[0, 0], K_in[1, 1]]
        npv = (prev_pts.astype(np.float64) - [K_in[0, 2], K_in[1, 2]]) / [K_in[0, 0], K_in[1, 1]]
        und_cur = np.stack([K_out[0, 0] * nc[:, 0] + K_out[0, 2], K_out[1, 1] * nc[:, 1] + K_out[1, 2]], axis=1)
        und_prev = npv
    img = und_cur.astype(np.float32).astype(np.float64)          # Point2f
    pn = und_prev.astype(np.float32).astype(np.float64)          # Point2f
    s = np.array([rng.uniform() for _ in range(n)])              # :345
    obj = np.stack([pn[:, 0] * s, pn[:, 1] * s, s], axis=1)      # :346-350
    f, cx, cy = float(K_out[0, 0]), float(K_out[0, 2]), float(K_out[1, 2])
    model_points, thresh2, confidence = 5, np.float32(64.0), 0.99
    niters, best_count, best, best_mask = 100, 0, None, None
    it = 0
    while it < niters:
        it += 1
        idx = []
        while len(idx) < model_points:
            c = rng.below(n)
            if c not in idx:
                idx.append(c)
        m = _solve_pnp_lm(obj[idx], img[idx], f, cx, cy, np.eye(3), np.zeros(3), 10)
        if m is None:
            continue
        with np.errstate(all="ignore"):
            e = _pnp_project(m[0], m[1], obj, f, cx, cy) - img
            err = (e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]).astype(np.float32)
            mask = err <= thresh2                                 # NaN compares false
        good = int(mask.sum())
        if good > max(best_count, model_points - 1):
            best_count, best, best_mask = good, m, mask
            niters = _ransac_update_iters(confidence, (n - good) / n, model_points, niters)
    if best_count <= 0:
        return np.eye(3), 0
    inl = np.nonzero(best_mask)[0]
    ref = _solve_pnp_lm(obj[inl], img[inl], f, cx, cy, best[0], best[1], 20)
    if ref is not None:
        best = ref
    return rodrigues(rodrigues_inv(best[0])), best_count


def project_to_output(points, K_in, K_out, R_warp, in_fish=True, out_fish=False):
    """Where the warp with rotation R_warp (the matrix handed to the map) sends input pixels: the inverse of the map --
    input pixel -> ray (input projection) -> R_warp^T -> output projection.  Returns integer pixel centres (rounded half
    to even) and a validity mask (ray in front of the output camera).  For the `debug` overlay."""
    pts = np.asarray(points, np.float64).reshape(-1, 2)
    a, b = (pts[:, 0] - K_in[0, 2]) / K_in[0, 0], (pts[:, 1] - K_in[1, 2]) / K_in[1, 1]
    if in_fish:
        th = np.hypot(a, b)
        sc = np.where(th > 0, np.sin(th) / np.where(th > 0, th, 1), 1.0)
        ray = np.stack([a * sc, b * sc, np.cos(th)], -1)
    else:
        ray = np.stack([a, b, np.ones_like(a)], -1)
    o = ray @ np.asarray(R_warp)                    # rows: R^T ray
    ok = o[:, 2] > 0
    with np.errstate(all="ignore"):
        u, v = o[:, 0] / o[:, 2], o[:, 1] / o[:, 2]
        if out_fish:
            r = np.hypot(o[:, 0], o[:, 1])
            sc = np.where(r > 0, np.arctan2(r, o[:, 2]) / np.where(r > 0, r, 1), 1.0)
            u, v = o[:, 0] * sc, o[:, 1] * sc
    c = np.stack([np.rint(K_out[0, 2] + u * K_out[0, 0]), np.rint(K_out[1, 2] + v * K_out[1, 1])], -1)
    return c[ok].astype(np.int64), ok


def draw_markers(img, centres, half, colour):
    """Filled (2*half+1)^2 squares, clipped.  img: (h, w, 3) BGR or (h, w) plane (modified in place)."""
    h, w = img.shape[:2]
    for x, y in centres:
        x0, x1, y0, y1 = max(x - half, 0), min(x + half + 1, w), max(y - half, 0), min(y + half + 1, h)
        if x0 < x1 and y0 < y1:
            img[y0:y1, x0:x1] = colour
    return img


class WarpStateMachine:
    """consume_frame / pull_frame control flow, FrameSourceWarp.cpp:397-476, with the pixel and
    estimation steps injected so the same state machine can be driven by oracle or by recorded
    product outputs.

    source: iterator of packed NV12 arrays.
    find_corners(gray) -> (n,2) float32                       (:228-240)
    track(prev_gray, gray, corners) -> (prev_pts, cur_pts)     (:242-270, status-filtered)
    estimate(prev_pts, cur_pts) -> (R 3x3, n_inliers)          (:316-375)
    warp(nv12, R) -> output frame                              (:272-314; cvtColor folded in)
    """

    def __init__(self, source, smooth_radius, find_corners, track, estimate, warp):
        self.source = iter(source)
        self.r = smooth_radius
        self.find_corners, self.track, self.estimate, self.warp = find_corners, track, estimate, warp
        self.filter = RotationFilter(smooth_radius)
        self.measured = np.eye(3)
        self.frame_index = 0
        self.last_key = -1
        self.last_gray = None
        self.corners = None
        self.last_rot = None
        self.frames = []
        self.rots = []
        self.log = []  # per consumed frame: dict(key=bool, n_tracked, inliers, R)

    def consume_frame(self, nv12):
        h = nv12.shape[0] * 2 // 3
        gray = nv12[:h]
        if self.last_key == -1:
            self.last_key = self.frame_index
            self.corners = self.find_corners(gray)
        else:
            key = False
            if self.frame_index - self.last_key > 20 or len(self.corners) < 150:
                self.last_key = self.frame_index - 1
                self.corners = self.find_corners(self.last_gray)
                key = True
            prev_pts, cur_pts = self.track(self.last_gray, gray, self.corners)
            self.corners = cur_pts
            R, inl = self.estimate(prev_pts, cur_pts)
            if inl < 40:
                R = np.eye(3) if self.last_rot is None else self.last_rot
            self.last_rot = R
            self.measured = R @ self.measured
            self.filter.add(self.measured)
            self.frames.append(nv12)
            self.rots.append(self.measured.copy())
            self.log.append(dict(key=key, n_tracked=len(cur_pts), inliers=inl, R=R))
        self.last_gray = gray
        self.frame_index += 1

    def pull_frame(self):
        """Returns the warped frame, or None at end of stream (the reference throws EOF)."""
        while len(self.frames) <= self.r:
            try:
                nv12 = next(self.source)
            except StopIteration:
                self.filter.add(self.measured)  # :459
                break
            self.consume_frame(nv12)
        if not self.frames:
            return None
        frame, measured = self.frames.pop(0), self.rots.pop(0)
        corrected = self.filter.filter()
        correction = corrected @ np.linalg.inv(measured)
        return self.warp(frame, np.linalg.inv(correction))
