// vstab_motion.cpp -- rotation estimation (guess_camera_rotation, FrameSourceWarp.cpp:316-375)
// and trajectory smoothing (gram_sg::RotationFilter, :212,444,459,471) on the host, fp64.
// The data is <= 200 points / one 3x3 matrix per frame, exactly as small as in the reference,
// which also runs these steps on the host.
#include "vstab_motion.hpp"
#include "vstab_internal.hpp"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>

namespace vstab {

Mat3 rodrigues(const double rv[3]) {
    const double th = std::sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    if (th < DBL_EPSILON) return Mat3::identity();
    const double k[3] = {rv[0] / th, rv[1] / th, rv[2] / th};
    const double c = std::cos(th), s = std::sin(th), c1 = 1 - c;
    Mat3 R;
    R(0, 0) = c + c1 * k[0] * k[0], R(0, 1) = c1 * k[0] * k[1] - s * k[2], R(0, 2) = c1 * k[0] * k[2] + s * k[1];
    R(1, 0) = c1 * k[0] * k[1] + s * k[2], R(1, 1) = c + c1 * k[1] * k[1], R(1, 2) = c1 * k[1] * k[2] - s * k[0];
    R(2, 0) = c1 * k[0] * k[2] - s * k[1], R(2, 1) = c1 * k[1] * k[2] + s * k[0], R(2, 2) = c + c1 * k[2] * k[2];
    return R;
}

void rodrigues_inv(const Mat3 &R, double rv[3]) {
    const double cs = std::min(1.0, std::max(-1.0, (R(0, 0) + R(1, 1) + R(2, 2) - 1) * 0.5));
    const double th = std::acos(cs);
    const double ax[3] = {R(2, 1) - R(1, 2), R(0, 2) - R(2, 0), R(1, 0) - R(0, 1)};
    const double sn = 0.5 * std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    if (sn < 1e-12) {
        if (cs > 0) {
            rv[0] = rv[1] = rv[2] = 0;
        } else {  // rotation by pi: axis from the diagonal
            double v[3] = {std::sqrt(std::max(0.0, (R(0, 0) + 1) * 0.5)), std::sqrt(std::max(0.0, (R(1, 1) + 1) * 0.5)),
                           std::sqrt(std::max(0.0, (R(2, 2) + 1) * 0.5))};
            if (R(0, 1) < 0) v[1] = -v[1];
            if (R(0, 2) < 0) v[2] = -v[2];
            for (int i = 0; i < 3; i++) rv[i] = v[i] * th;
        }
        return;
    }
    const double f = th / (2 * sn);
    for (int i = 0; i < 3; i++) rv[i] = ax[i] * f;
}

// ---------------------------------------------------------------------------------------------
// Perspective-n-point by Levenberg-Marquardt on the reprojection error.  The reference calls
// OpenCV's solvePnPRansac (EPnP on 5-point samples + SOLVEPNP_ITERATIVE refit); that library is
// not available, and the reference's result is not reproducible anyway (un-seeded rand() feeds
// the depths, F6), so parity for this step is by ground truth, not bitwise.  Inter-frame motion is
// small, so identity is always a good starting point for the minimal solver.
// ---------------------------------------------------------------------------------------------
struct Pose {
    Mat3 R = Mat3::identity();
    double t[3] = {0, 0, 0};
};

static inline void project(const Pose &p, const double X[3], double f, double cx, double cy, double out[2], double Y[3]) {
    for (int r = 0; r < 3; r++) Y[r] = p.R(r, 0) * X[0] + p.R(r, 1) * X[1] + p.R(r, 2) * X[2] + p.t[r];
    out[0] = f * Y[0] / Y[2] + cx, out[1] = f * Y[1] / Y[2] + cy;
}

static bool solve6(double A[36], double b[6]) {  // Gaussian elimination with partial pivoting, in place
    for (int c = 0; c < 6; c++) {
        int piv = c;
        for (int r = c + 1; r < 6; r++)
            if (std::fabs(A[r * 6 + c]) > std::fabs(A[piv * 6 + c])) piv = r;
        if (std::fabs(A[piv * 6 + c]) < 1e-300) return false;
        if (piv != c) {
            for (int k = 0; k < 6; k++) std::swap(A[c * 6 + k], A[piv * 6 + k]);
            std::swap(b[c], b[piv]);
        }
        for (int r = c + 1; r < 6; r++) {
            const double m = A[r * 6 + c] / A[c * 6 + c];
            for (int k = c; k < 6; k++) A[r * 6 + k] -= m * A[c * 6 + k];
            b[r] -= m * b[c];
        }
    }
    for (int c = 5; c >= 0; c--) {
        for (int k = c + 1; k < 6; k++) b[c] -= A[c * 6 + k] * b[k];
        b[c] /= A[c * 6 + c];
    }
    return true;
}

static double reproj_cost(const Pose &p, const double *obj, const float *img, const int *idx, int n, double f, double cx,
                          double cy) {
    double cost = 0;
    for (int i = 0; i < n; i++) {
        const int k = idx ? idx[i] : i;
        double u[2], Y[3];
        project(p, obj + 3 * k, f, cx, cy, u, Y);
        const double ex = u[0] - img[2 * k], ey = u[1] - img[2 * k + 1];
        cost += ex * ex + ey * ey;
    }
    return cost;
}

static bool solve_pnp_lm(const double *obj, const float *img, const int *idx, int n, double f, double cx, double cy,
                         Pose &pose, int max_iter) {
    double lambda = 1e-3;
    double cost = reproj_cost(pose, obj, img, idx, n, f, cx, cy);
    if (!std::isfinite(cost)) return false;
    for (int it = 0; it < max_iter; it++) {
        // normal equations: upper triangle of J^T J (21 entries) and J^T r, with the closed-form Jacobian
        // of u = f*x/z + cx, v = f*y/z + cy with respect to a left rotation increment w and translation t:
        //   dY/dw = -[R X]x , dY/dt = I
        double H[6][6] = {{0}}, g[6] = {0};
        for (int i = 0; i < n; i++) {
            const int k = idx ? idx[i] : i;
            const double *X = obj + 3 * k;
            const double x = pose.R(0, 0) * X[0] + pose.R(0, 1) * X[1] + pose.R(0, 2) * X[2];
            const double y = pose.R(1, 0) * X[0] + pose.R(1, 1) * X[1] + pose.R(1, 2) * X[2];
            const double z = pose.R(2, 0) * X[0] + pose.R(2, 1) * X[1] + pose.R(2, 2) * X[2];  // R X
            const double Yx = x + pose.t[0], Yy = y + pose.t[1], Yz = z + pose.t[2];
            const double iz = 1.0 / Yz, fz = f * iz, xz = Yx * iz, yz = Yy * iz;
            const double eu = fz * Yx + cx - img[2 * k], ev = fz * Yy + cy - img[2 * k + 1];
            const double Ju[6] = {-fz * xz * y, fz * (z + xz * x), -fz * y, fz, 0.0, -fz * xz};
            const double Jv[6] = {-fz * (z + yz * y), fz * yz * x, fz * x, 0.0, fz, -fz * yz};
            for (int c = 0; c < 6; c++) {
                g[c] += Ju[c] * eu + Jv[c] * ev;
                for (int d = c; d < 6; d++) H[c][d] += Ju[c] * Ju[d] + Jv[c] * Jv[d];
            }
        }
        bool improved = false;
        for (int tries = 0; tries < 8 && !improved; tries++) {
            double A[36], b[6];
            for (int c = 0; c < 6; c++) {
                for (int d = 0; d < 6; d++) A[c * 6 + d] = c <= d ? H[c][d] : H[d][c];
                A[c * 6 + c] += lambda * (H[c][c] + 1e-12), b[c] = -g[c];
            }
            if (!solve6(A, b)) {
                lambda *= 10;
                continue;
            }
            Pose cand;
            cand.R = rodrigues(b) * pose.R;
            for (int c = 0; c < 3; c++) cand.t[c] = pose.t[c] + b[3 + c];
            const double c2 = reproj_cost(cand, obj, img, idx, n, f, cx, cy);
            if (std::isfinite(c2) && c2 < cost) {
                const double rel = (cost - c2) / std::max(cost, 1e-300);
                pose = cand, cost = c2, lambda = std::max(lambda * 0.1, 1e-12), improved = true;
                if (rel < 1e-9 || cost < 1e-20) return true;
            } else {
                lambda *= 10;
            }
        }
        if (!improved) break;
    }
    return true;
}

// cv::RANSACUpdateNumIters (OpenCV calib3d ptsetreg.cpp)
static int ransac_update_iters(double p, double ep, int model_points, int max_iters) {
    p = std::min(std::max(p, 0.0), 1.0), ep = std::min(std::max(ep, 0.0), 1.0);
    double num = std::max(1.0 - p, DBL_MIN), denom = 1.0 - std::pow(1.0 - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num), denom = std::log(denom);
    return (denom >= 0 || -num >= max_iters * (-denom)) ? max_iters : (int)std::nearbyint(num / denom);
}

int estimate_rotation(const float *prev, const float *cur, int n, const Mat3 &Kin, const Mat3 &Kout, Pcg32 &rng, Mat3 &R,
                      bool in_fish) {
    R = Mat3::identity();
    if (n < 5) return 0;  // solvePnPRansac needs >= model size; the reference maps failure to (I, 0) (:367-371)
    static thread_local std::vector<double> tmp, und, obj;
    static thread_local std::vector<float> img;
    static thread_local std::vector<uint8_t> best_mask, mask;
    static thread_local std::vector<int> inl;
    tmp.resize(2 * (size_t)n), und.resize(2 * (size_t)n), obj.resize(3 * (size_t)n), img.resize(2 * (size_t)n);
    // :322-330 current points -> output-camera pixels (R = I, P = output matrix), stored as Point2f
    for (int i = 0; i < 2 * n; i++) tmp[i] = cur[i];
    fisheye_undistort(tmp.data(), n, Kin, Kout, und.data(), in_fish);
    for (int i = 0; i < 2 * n; i++) img[i] = (float)und[i];
    // :333-338 previous points -> normalised identity-camera coordinates, stored as Point2f
    for (int i = 0; i < 2 * n; i++) tmp[i] = prev[i];
    fisheye_undistort(tmp.data(), n, Kin, Mat3::identity(), und.data(), in_fish);
    for (int i = 0; i < n; i++) {
        const double s = rng.uniform();  // :345 random depth ("prevents the detection of translations")
        const float px = (float)und[2 * i], py = (float)und[2 * i + 1];
        obj[3 * i] = px * s, obj[3 * i + 1] = py * s, obj[3 * i + 2] = s;
    }
    const double f = Kout(0, 0), cx = Kout(0, 2), cy = Kout(1, 2);
    // :354-366 solvePnPRansac(100 iterations, 8 px, 0.99), minimal sample 5
    const int model_points = 5;
    const double thresh2 = 8.0 * 8.0, confidence = 0.99;
    int niters = 100, best_count = 0;
    Pose best;
    best_mask.assign(n, 0), mask.assign(n, 0);
    for (int iter = 0; iter < niters; iter++) {
        int idx[5];
        for (int k = 0; k < model_points;) {
            const int c = (int)rng.below((uint32_t)n);
            bool dup = false;
            for (int j = 0; j < k; j++) dup |= idx[j] == c;
            if (!dup) idx[k++] = c;
        }
        Pose pose;
        if (!solve_pnp_lm(obj.data(), img.data(), idx, model_points, f, cx, cy, pose, 10)) continue;
        int good = 0;
        for (int i = 0; i < n; i++) {
            double u[2], Y[3];
            project(pose, obj.data() + 3 * i, f, cx, cy, u, Y);
            const double ex = u[0] - img[2 * i], ey = u[1] - img[2 * i + 1];
            const float err = (float)(ex * ex + ey * ey);
            mask[i] = err <= (float)thresh2;  // NaN compares false
            good += mask[i];
        }
        if (good > std::max(best_count, model_points - 1)) {
            best_count = good, best = pose, best_mask = mask;
            niters = ransac_update_iters(confidence, (double)(n - good) / n, model_points, niters);
        }
    }
    if (best_count <= 0) return 0;
    // refit on all inliers (SOLVEPNP_ITERATIVE in the reference), starting from the best sample model
    inl.clear();
    for (int i = 0; i < n; i++)
        if (best_mask[i]) inl.push_back(i);
    Pose refined = best;
    if (solve_pnp_lm(obj.data(), img.data(), inl.data(), (int)inl.size(), f, cx, cy, refined, 20)) best = refined;
    // :373 Rodrigues(rvec) -- re-orthonormalise through the rotation vector as the reference does
    double rv[3];
    rodrigues_inv(best.R, rv);
    R = rodrigues(rv);
    return best_count;
}

// ---------------------------------------------------------------------------------------------
// Savitzky-Golay rotation filter
// ---------------------------------------------------------------------------------------------
std::vector<double> sg_weights(int m) {
    // closed form of Gorry's Gram-polynomial weights for n = 2, t = 0, s = 0
    std::vector<double> w(2 * (size_t)m + 1);
    const double den = (2.0 * m - 1) * (2.0 * m + 1) * (2.0 * m + 3);
    for (int i = -m; i <= m; i++) w[i + m] = 3.0 * (3.0 * m * m + 3.0 * m - 1 - 5.0 * i * i) / den;
    if (m == 0) w[0] = 1.0;
    return w;
}

Mat3 polar_orthogonal(const Mat3 &M) {
    // Newton iteration X <- (X + X^-T)/2 with determinant scaling converges to the orthogonal
    // polar factor U*V^T for any non-singular M (including det < 0, where it is a reflection --
    // exactly what JacobiSVD's U*V^T gives, "no determinant fix").
    Mat3 X = M;
    const double d0 = X.det();
    if (!(std::fabs(d0) > 1e-300) || !std::isfinite(d0)) return Mat3::identity();
    for (int it = 0; it < 100; it++) {
        const double d = X.det();
        if (!(std::fabs(d) > 1e-300)) break;
        const Mat3 Xit = X.inv().t();
        const double g = std::pow(std::fabs(d), -1.0 / 3.0);  // scaling accelerates far-from-orthogonal starts
        Mat3 Y;
        double diff = 0;
        for (int i = 0; i < 9; i++) {
            Y.m[i] = 0.5 * (g * X.m[i] + Xit.m[i] / g);
            diff = std::max(diff, std::fabs(Y.m[i] - X.m[i]));
        }
        X = Y;
        if (diff < 1e-15) break;
    }
    return X;
}

RotationFilterSG::RotationFilterSG(int m) : m_(m), w_(sg_weights(m)), ring_(2 * (size_t)m + 1, Mat3::zero()) {}

void RotationFilterSG::add(const Mat3 &R) {
    ring_[head_] = R;  // overwrite the oldest; the new oldest is the next slot
    head_ = (head_ + 1) % ring_.size();
}

Mat3 RotationFilterSG::filter() const {
    Mat3 acc = Mat3::zero();
    for (size_t i = 0; i < ring_.size(); i++) {
        const Mat3 &r = ring_[(head_ + i) % ring_.size()];
        for (int k = 0; k < 9; k++) acc.m[k] += w_[i] * r.m[k];
    }
    return polar_orthogonal(acc);
}

// ---------------------------------------------------------------------------------------------
// Kalman mode
// ---------------------------------------------------------------------------------------------
double RotationFilterKalman::Axis::step(double z) {
    const double Q = 1e-5, Rn = 1e-1;
    // predict: x = F x, P = F P F^T + Q
    const double x0 = x[0] + x[1], x1 = x[1];
    const double p00 = P[0] + P[1] + P[2] + P[3] + Q, p01 = P[1] + P[3], p10 = P[2] + P[3], p11 = P[3] + Q;
    // correct with H = [1 0]
    const double S = p00 + Rn, k0 = p00 / S, k1 = p10 / S, y = z - x0;
    x[0] = x0 + k0 * y, x[1] = x1 + k1 * y;
    P[0] = (1 - k0) * p00, P[1] = (1 - k0) * p01, P[2] = p10 - k1 * p00, P[3] = p11 - k1 * p01;
    return x[0];
}

RotationFilterKalman::RotationFilterKalman() {}

Mat3 RotationFilterKalman::update(const Mat3 &measured) {
    double rv[3], out[3];
    rodrigues_inv(measured, rv);
    for (int i = 0; i < 3; i++) out[i] = ax_[i].step(rv[i]);
    return rodrigues(out);
}

}  // namespace vstab

// ---------------------------------------------------------------------------------------------
// Gyro samples -> rotations (the step gpmf.cpp:5-11 / AvFrameSourceFileVaapi.cpp:121-123 stub out): ordered product of
// exponential maps of rate * overlap, later samples on the left (include/vstab.h).
// ---------------------------------------------------------------------------------------------
namespace vstab {
static Mat3 gyro_span(const vstab_gyro_sample *s, int n, double scale, double a, double b) {
    Mat3 R = Mat3::identity();
    for (int i = 0; i < n; i++) {
        const double lo = s[i].start_ts > a ? s[i].start_ts : a, hi = s[i].end_ts < b ? s[i].end_ts : b;
        if (!(hi > lo)) continue;
        const double dt = (hi - lo) * scale;
        const double rv[3] = {s[i].pitch * dt, s[i].yaw * dt, s[i].roll * dt};
        R = rodrigues(rv) * R;
    }
    return R;
}
}  // namespace vstab

extern "C" vstab_status vstab_gyro_integrate(const vstab_gyro_sample *samples, int n, double rate_scale, double t_prev_first_row,
                                             double t_first_row, double t_last_row, double R_delta[9], double R_readout[9]) {
    using namespace vstab;
    if (n < 0 || (n > 0 && !samples)) return fail(VSTAB_ERR_INVALID, "vstab_gyro_integrate: null samples");
    if (!(rate_scale == rate_scale) || !(t_prev_first_row <= t_first_row) || !(t_first_row <= t_last_row))
        return fail(VSTAB_ERR_INVALID, "vstab_gyro_integrate: need t_prev_first_row <= t_first_row <= t_last_row");
    for (int i = 0; i < n; i++) {
        if (!(samples[i].end_ts >= samples[i].start_ts) || (i > 0 && !(samples[i].start_ts >= samples[i - 1].start_ts)))
            return fail(VSTAB_ERR_INVALID, "vstab_gyro_integrate: samples must be ordered by start_ts with end_ts >= start_ts");
    }
    if (R_delta) {
        const Mat3 R = gyro_span(samples, n, rate_scale, t_prev_first_row, t_first_row);
        for (int k = 0; k < 9; k++) R_delta[k] = R.m[k];
    }
    if (R_readout) {
        const Mat3 R = gyro_span(samples, n, rate_scale, t_first_row, t_last_row);
        for (int k = 0; k < 9; k++) R_readout[k] = R.m[k];
    }
    return VSTAB_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// GPMF payload -> gyro samples: what opencv/gpmf.cpp:33-114 sketches with gpmf-parser (commented out there) and
// AvFrameSourceFileVaapi.cpp:121-123 leaves as "TODO process GPMF packet".  gpmf-parser is not in the image; the payload format
// is GoPro's published KLV: every item is a FourCC key, a type character, the size of one sample structure (1 byte), a
// repeat count (2 bytes, big-endian) and repeat * size bytes of big-endian data, padded to a multiple of four; type 0 nests
// (DEVC device > STRM stream > items).  Inside a STRM, SCAL holds the divisor(s) that turn the raw GYRO integers into rad/s.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct GpmfItem {
    uint32_t key;
    char type;
    uint32_t size, repeat;
    const uint8_t *data;
    size_t bytes;  // repeat * size (unpadded)
};
constexpr uint32_t fourcc(char a, char b, char c, char d) { return (uint32_t)(uint8_t)a << 24 | (uint32_t)(uint8_t)b << 16 | (uint32_t)(uint8_t)c << 8 | (uint32_t)(uint8_t)d; }

// size of one element of a simple numeric type, 0 for anything else
int gpmf_elem_size(char t) {
    switch (t) {
        case 'b': case 'B': return 1;
        case 's': case 'S': return 2;
        case 'l': case 'L': case 'f': return 4;
        case 'd': case 'j': case 'J': return 8;
        default: return 0;
    }
}
double gpmf_elem(const uint8_t *p, char t) {
    uint64_t v = 0;
    const int n = gpmf_elem_size(t);
    for (int i = 0; i < n; i++) v = v << 8 | p[i];
    switch (t) {
        case 'b': return (double)(int8_t)v;
        case 'B': return (double)(uint8_t)v;
        case 's': return (double)(int16_t)v;
        case 'S': return (double)(uint16_t)v;
        case 'l': return (double)(int32_t)v;
        case 'L': return (double)(uint32_t)v;
        case 'j': return (double)(int64_t)v;
        case 'J': return (double)v;
        case 'f': { const uint32_t u = (uint32_t)v; float f; std::memcpy(&f, &u, 4); return (double)f; }
        default: { double d; std::memcpy(&d, &v, 8); return d; }  // 'd'
    }
}

struct GpmfGyroSink {
    double ts, dur;
    vstab_gyro_sample *out;
    int cap, n = 0;
    std::string err;
};

// one level of the tree: the items of [p, p + n); scal = the divisors of the enclosing STRM seen so far
bool gpmf_walk(const uint8_t *p, size_t n, int depth, GpmfGyroSink &sink) {
    if (depth > 8) return sink.err = "nesting deeper than 8 levels", false;
    double scal[3] = {1, 1, 1};
    size_t pos = 0;
    while (pos < n) {
        if (n - pos < 4) return sink.err = "truncated item header", false;
        GpmfItem it;
        it.key = (uint32_t)p[pos] << 24 | (uint32_t)p[pos + 1] << 16 | (uint32_t)p[pos + 2] << 8 | p[pos + 3];
        if (it.key == 0) break;  // zero padding at the end of a payload
        if (n - pos < 8) return sink.err = "truncated item header", false;
        it.type = (char)p[pos + 4], it.size = p[pos + 5], it.repeat = (uint32_t)p[pos + 6] << 8 | p[pos + 7];
        it.bytes = (size_t)it.size * it.repeat;
        const size_t padded = (it.bytes + 3) & ~(size_t)3;
        if (padded > n - pos - 8) return sink.err = "item longer than its container", false;
        it.data = p + pos + 8;
        if (it.type == 0) {  // nested container
            if (!gpmf_walk(it.data, it.bytes, depth + 1, sink)) return false;
        } else if (it.key == fourcc('S', 'C', 'A', 'L')) {
            const int es = gpmf_elem_size(it.type);
            if (es == 0 || it.size == 0 || it.size % es) return sink.err = "SCAL of a type that is not a plain number", false;
            const uint32_t count = it.size / es * it.repeat;  // one divisor for all elements, or one per element
            if (count != 1 && count != 3) {
                scal[0] = scal[1] = scal[2] = 1;  // some other stream's scale (GPS5 has five): not a gyro's
            } else {
                for (int k = 0; k < 3; k++) scal[k] = gpmf_elem(it.data + (count == 3 ? k * es : 0), it.type);
                for (int k = 0; k < 3; k++)
                    if (!(std::isfinite(scal[k]) && scal[k] != 0)) return sink.err = "SCAL of zero (or not a finite number)", false;
            }
        } else if (it.key == fourcc('G', 'Y', 'R', 'O')) {
            const int es = gpmf_elem_size(it.type);
            if (es == 0 || it.size != 3u * es) return sink.err = "Unexpected number of elements for GYRO data", false;  // gpmf.cpp:88-92
            for (uint32_t s = 0; s < it.repeat; s++) {
                if (sink.n < sink.cap) {
                    vstab_gyro_sample &g = sink.out[sink.n];
                    const uint8_t *e = it.data + (size_t)s * it.size;
                    // gpmf.cpp:95-101: the packet's time span dealt evenly to its samples; elements 0, 1, 2 -> roll, pitch, yaw
                    // (GoPro documents the order Z, X, Y for HERO5 and later: about the optical axis, to the right, downwards)
                    g.start_ts = sink.ts + sink.dur * s / it.repeat;
                    g.end_ts = g.start_ts + sink.dur / it.repeat;
                    g.roll = gpmf_elem(e, it.type) / scal[0], g.pitch = gpmf_elem(e + es, it.type) / scal[1], g.yaw = gpmf_elem(e + 2 * es, it.type) / scal[2];
                }
                if (sink.n == INT_MAX) return sink.err = "more samples than an int counts", false;
                sink.n++;
            }
        }
        pos += 8 + padded;
    }
    return true;
}

}  // namespace

using vstab::fail;
extern "C" vstab_status vstab_gpmf_parse_gyro(const void *payload, size_t n, double pkt_ts, double pkt_dur, vstab_gyro_sample *out, int cap, int *n_out) {
    if (n_out) *n_out = 0;
    if (!payload || !n_out || cap < 0 || (cap > 0 && !out)) return fail(VSTAB_ERR_INVALID, "vstab_gpmf_parse_gyro: null argument");
    if (!std::isfinite(pkt_ts) || !std::isfinite(pkt_dur) || pkt_dur < 0)
        return fail(VSTAB_ERR_INVALID, "vstab_gpmf_parse_gyro: packet time stamp and duration must be finite, the duration >= 0");
    GpmfGyroSink sink{pkt_ts, pkt_dur, out, cap};
    if (!gpmf_walk(static_cast<const uint8_t *>(payload), n, 0, sink)) return fail(VSTAB_ERR_INVALID, "vstab_gpmf_parse_gyro: " + sink.err);
    *n_out = sink.n;
    return VSTAB_OK;
}
