// vstab_geometry.cpp -- host-side camera geometry (fp64) behind the C ABI.
// Replaces get_preset_camera / get_output_camera (FrameSourceWarp.cpp:27-165) and the
// cv::fisheye::undistortPoints calls (:93, :322, :333) for the zero-distortion case the
// reference uses.  These run once per clip / on <= 200 points per frame, so they stay on the
// host exactly as in the reference.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "vstab_geometry.hpp"
#include "vstab_internal.hpp"

namespace vstab {

static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }

static thread_local LaunchEvents g_launch_events;
void set_launch_events(hipEvent_t start, hipEvent_t stop) { g_launch_events = {start, stop}; }
LaunchEvents take_launch_events() {
    const LaunchEvents e = g_launch_events;
    g_launch_events = LaunchEvents();
    return e;
}
bool launch_events_pending() { return g_launch_events.start != nullptr; }

// FrameSourceWarp.cpp:22-25: the published FOVs are stored in `const int`, dropping the fraction.
static const int FOV_H_43W = (int)122.6, FOV_V_43W = (int)94.4, FOV_H_169W = (int)118.2, FOV_V_169W = (int)69.5;

bool preset_camera(int preset, int w, int h, Mat3 &K) {
    K = Mat3::identity();
    K(0, 2) = (w - 1.) / 2;  // :31-32 principal point at the centre by default
    K(1, 2) = (h - 1.) / 2;
    switch (preset) {
        case VSTAB_GOPRO_H4B_WIDE43_PUBLISHED:
            K(0, 0) = w / (FOV_H_43W * M_PI / 180);
            K(1, 1) = h / (FOV_V_43W * M_PI / 180);
            break;
        case VSTAB_GOPRO_H4B_WIDE169_PUBLISHED:
            K(0, 0) = w / (FOV_H_169W * M_PI / 180);
            K(1, 1) = h / (FOV_V_169W * M_PI / 180);
            break;
        case VSTAB_GOPRO_H4B_WIDE43_MEASURED:  // :50-56 (fx AND fy scale with height)
            K(0, 2) = 967.37 * w / 1920, K(1, 2) = 711.07 * h / 1440;
            K(0, 0) = 942.96 * h / 1440, K(1, 1) = 942.53 * h / 1440;
            break;
        case VSTAB_GOPRO_H4B_WIDE43_MEASURED_STABILISATION:
            K(0, 2) = 965.90 * w / 1920, K(1, 2) = 712.94 * h / 1440;
            K(0, 0) = 1045.58 * h / 1440, K(1, 1) = 1045.64 * h / 1440;
            break;
        case VSTAB_GOPRO_H4B_WIDE169_MEASURED:
            K(0, 2) = 1361.80 * w / 2704, K(1, 2) = 745.19 * h / 1520;
            K(0, 0) = 1392.49 * h / 1520, K(1, 1) = 1383.47 * h / 1520;
            break;
        case VSTAB_GOPRO_H4B_WIDE169_MEASURED_STABILISATION:
            K(0, 2) = 1357.49 * w / 2704, K(1, 2) = 736.74 * h / 1520;
            K(0, 0) = 1626.67 * h / 1520, K(1, 1) = 1619.46 * h / 1520;
            break;
        default:
            return false;
    }
    return true;
}

// OpenCV 4.5 fisheye::undistortPoints with D = 0: theta_d clipped to pi/2, scale = tan(theta)/theta.
void fisheye_undistort(const double *pts, int n, const Mat3 &K, const Mat3 &RR, double *out, bool fish) {
    for (int i = 0; i < n; i++) {
        const double pwx = (pts[2 * i] - K(0, 2)) / K(0, 0), pwy = (pts[2 * i + 1] - K(1, 2)) / K(1, 1);
        double theta_d = std::sqrt(pwx * pwx + pwy * pwy);
        theta_d = std::min(std::max(-M_PI / 2., theta_d), M_PI / 2.);
        double scale = 0.0;
        if (std::fabs(theta_d) > 1e-8) scale = std::tan(theta_d) / theta_d;
        if (!fish) scale = 1.0;
        const double ux = pwx * scale, uy = pwy * scale;
        const double x = RR(0, 0) * ux + RR(0, 1) * uy + RR(0, 2);
        const double y = RR(1, 0) * ux + RR(1, 1) * uy + RR(1, 2);
        const double z = RR(2, 0) * ux + RR(2, 1) * uy + RR(2, 2);
        out[2 * i] = x / z, out[2 * i + 1] = y / z;
    }
}

bool lens_camera(int projection, double dfov_deg, int w, int h, double cx, double cy, Mat3 &K) {
    if (w <= 0 || h <= 0 || !(dfov_deg > 0)) return false;
    const double half_diag = 0.5 * std::sqrt((double)w * w + (double)h * h), half_fov = 0.5 * dfov_deg * M_PI / 180.0;
    double f;
    if (projection == VSTAB_PROJ_RECT) {
        if (!(dfov_deg < 180)) return false;
        f = half_diag / std::tan(half_fov);
    } else if (projection == VSTAB_PROJ_FISH) {
        if (!(dfov_deg < 360)) return false;
        f = half_diag / half_fov;
    } else {
        return false;
    }
    K = Mat3::identity();
    K(0, 0) = K(1, 1) = f;
    K(0, 2) = cx < 0 ? 0.5 * w : cx, K(1, 2) = cy < 0 ? 0.5 * h : cy;
    return true;
}

static int cv_round(double v) { return (int)std::nearbyint(v); }  // round half to even

void output_camera(const Mat3 &Kin, int w, int h, double scale, bool crop, double zoom, Mat3 &Kout, int &ow,
                   int &oh) {
    const double probes[16] = {0, 0, 0, h - 1., w - 1., 0, w - 1., h - 1.,                  // corners :96-99
                               Kin(0, 2), 0, w - 1., Kin(1, 2), Kin(0, 2), h - 1., 0, Kin(1, 2)};  // edge mid-points
    double ext[16];
    fisheye_undistort(probes, 8, Kin, Mat3::identity(), ext);
    const int start = crop ? 4 : 0;  // :119
    double max_x = ext[2 * start], min_x = max_x, max_y = ext[2 * start + 1], min_y = max_y;
    for (int i = start; i < 8; i++) {
        max_x = std::max(max_x, ext[2 * i]), min_x = std::min(min_x, ext[2 * i]);
        max_y = std::max(max_y, ext[2 * i + 1]), min_y = std::min(min_y, ext[2 * i + 1]);
    }
    const int idx = cv_round(w - 1.), idy = cv_round(h - 1.);  // :142 cv::Point (int)
    const double in_len = std::sqrt(1. * idx * idx + idy * idy);
    const int odx = cv_round(ext[6] - ext[0]), ody = cv_round(ext[7] - ext[1]);  // :146 cv::Point (int)
    const double out_len = std::sqrt(1. * odx * odx + ody * ody);
    scale *= in_len / out_len;  // :150
    Kout = Mat3::identity();
    Kout(0, 0) = scale, Kout(1, 1) = scale;
    Kout(0, 2) = scale * -min_x / zoom, Kout(1, 2) = scale * -min_y / zoom;
    ow = (int)(scale * (max_x - min_x) / zoom);  // :163 cv::Size(double,double) truncates
    oh = (int)(scale * (max_y - min_y) / zoom);
}

void map_params(const Mat3 &Kin, const Mat3 &Kout, const Mat3 &R, float p[17]) {
    p[0] = (float)Kin(0, 2), p[1] = (float)Kin(1, 2), p[2] = (float)Kin(0, 0), p[3] = (float)Kin(1, 1);
    p[4] = (float)Kout(0, 2), p[5] = (float)Kout(1, 2), p[6] = (float)Kout(0, 0), p[7] = (float)Kout(1, 1);
    for (int i = 0; i < 9; i++) p[8 + i] = (float)R.m[i];
}

}  // namespace vstab

using namespace vstab;

extern "C" {

const char *vstab_last_error(void) { return g_last_error.c_str(); }

const char *vstab_version(void) { return "vstab 0.1 gfx950"; }

int vstab_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
        return e == hipErrorNoDevice ? 0 : (int)VSTAB_ERR_DEVICE;
    }
    return n;
}

vstab_status vstab_get_preset_camera(int preset, int width, int height, double K[9]) {
    if (!K || width <= 0 || height <= 0) return fail(VSTAB_ERR_INVALID, "vstab_get_preset_camera: bad argument");
    Mat3 m;
    if (!preset_camera(preset, width, height, m)) return fail(VSTAB_ERR_INVALID, "vstab_get_preset_camera: unknown preset");
    std::memcpy(K, m.m, sizeof(m.m));
    return VSTAB_OK;
}

vstab_status vstab_get_output_camera(const double K_in[9], int width, int height, double scale, int crop_borders,
                                     double zoom, double K_out[9], int *out_width, int *out_height) {
    if (!K_in || !K_out || !out_width || !out_height || width <= 0 || height <= 0 || !(scale > 0) || !(zoom > 0))
        return fail(VSTAB_ERR_INVALID, "vstab_get_output_camera: bad argument");
    Mat3 ki, ko;
    std::memcpy(ki.m, K_in, sizeof(ki.m));
    output_camera(ki, width, height, scale, crop_borders != 0, zoom, ko, *out_width, *out_height);
    std::memcpy(K_out, ko.m, sizeof(ko.m));
    return VSTAB_OK;
}

vstab_status vstab_lens_camera(int projection, double dfov_deg, int width, int height, double cx, double cy, double K[9]) {
    Mat3 k;
    if (!K || !lens_camera(projection, dfov_deg, width, height, cx, cy, k))
        return fail(VSTAB_ERR_INVALID, "vstab_lens_camera: bad projection, field of view or size");
    std::memcpy(K, k.m, sizeof(k.m));
    return VSTAB_OK;
}

vstab_status vstab_fisheye_undistort_points(const double *pts, int n, const double K[9], const double *R,
                                            const double *P, double *out) {
    if (!pts || !K || !out || n < 0) return fail(VSTAB_ERR_INVALID, "vstab_fisheye_undistort_points: bad argument");
    Mat3 k, rr = Mat3::identity();
    std::memcpy(k.m, K, sizeof(k.m));
    if (R) std::memcpy(rr.m, R, sizeof(rr.m));
    if (P) {
        Mat3 p;
        std::memcpy(p.m, P, sizeof(p.m));
        rr = p * rr;
    }
    fisheye_undistort(pts, n, k, rr, out);
    return VSTAB_OK;
}

void vstab_map_params(const double K_in[9], const double K_out[9], const double R[9], float params[17]) {
    Mat3 a, b, c;
    std::memcpy(a.m, K_in, sizeof(a.m));
    std::memcpy(b.m, K_out, sizeof(b.m));
    std::memcpy(c.m, R, sizeof(c.m));
    map_params(a, b, c, params);
}

}  // extern "C"

extern "C" void vstab_time_next_launch(void *start_event, void *stop_event) {
    vstab::set_launch_events(static_cast<hipEvent_t>(start_event), static_cast<hipEvent_t>(stop_event));
}
