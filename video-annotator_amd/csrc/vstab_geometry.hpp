// vstab_geometry.hpp -- small fp64 3x3 helpers and the camera functions used by the pipeline.
#pragma once
#include <cmath>

namespace vstab {

struct Mat3 {
    double m[9];
    double &operator()(int r, int c) { return m[r * 3 + c]; }
    double operator()(int r, int c) const { return m[r * 3 + c]; }
    static Mat3 identity() { return {{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
    static Mat3 zero() { return {{0, 0, 0, 0, 0, 0, 0, 0, 0}}; }
    Mat3 operator*(const Mat3 &o) const {
        Mat3 r;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) r(i, j) = m[i * 3] * o(0, j) + m[i * 3 + 1] * o(1, j) + m[i * 3 + 2] * o(2, j);
        return r;
    }
    Mat3 t() const { return {{m[0], m[3], m[6], m[1], m[4], m[7], m[2], m[5], m[8]}}; }
    double det() const {
        return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    }
    // General inverse by cofactors (cv::Mat::inv() at FrameSourceWarp.cpp:472,475 is LU; the
    // matrices there are rotations so any exact method agrees to rounding).
    Mat3 inv() const {
        const double d = det();
        Mat3 r;
        r.m[0] = (m[4] * m[8] - m[5] * m[7]) / d, r.m[1] = (m[2] * m[7] - m[1] * m[8]) / d, r.m[2] = (m[1] * m[5] - m[2] * m[4]) / d;
        r.m[3] = (m[5] * m[6] - m[3] * m[8]) / d, r.m[4] = (m[0] * m[8] - m[2] * m[6]) / d, r.m[5] = (m[2] * m[3] - m[0] * m[5]) / d;
        r.m[6] = (m[3] * m[7] - m[4] * m[6]) / d, r.m[7] = (m[1] * m[6] - m[0] * m[7]) / d, r.m[8] = (m[0] * m[4] - m[1] * m[3]) / d;
        return r;
    }
};

bool preset_camera(int preset, int w, int h, Mat3 &K);
// fish = false: pinhole input, the points are only normalised and rotated
void fisheye_undistort(const double *pts, int n, const Mat3 &K, const Mat3 &RR, double *out, bool fish = true);
// camera matrix of a libdewobble-style lens (projection 0 rect / 1 fish, diagonal field of view in degrees)
bool lens_camera(int projection, double dfov_deg, int w, int h, double cx, double cy, Mat3 &K);
void output_camera(const Mat3 &Kin, int w, int h, double scale, bool crop, double zoom, Mat3 &Kout, int &ow, int &oh);
void map_params(const Mat3 &Kin, const Mat3 &Kout, const Mat3 &R, float p[17]);

}  // namespace vstab
