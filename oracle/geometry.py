"""Host-geometry part of the CPU oracle (numpy, fp64) -- TEST INFRASTRUCTURE ONLY.

File:line citations are into /root/reference/opencv/.
"""
import math

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
