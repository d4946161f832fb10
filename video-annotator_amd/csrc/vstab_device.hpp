// vstab_device.hpp -- device-side arithmetic shared by the HIP kernels (gfx950 only).
//
// Everything here must reproduce the reference arithmetic bit for bit, so this translation unit
// is compiled with -ffp-contract=off: a*b+c is two roundings unless written as __builtin_fmaf.
// hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt keeps `/` and sqrtf IEEE-correct.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vstab {

// ---------------------------------------------------------------------------------------------
// createMap.cl:13-50 arithmetic.  The 17 kernel arguments in argument order.
// ---------------------------------------------------------------------------------------------
struct MapParams {
    float icx, icy, ifx, ify;  // src_center_x/y, src_focal_x/y     (createMap.cl:4)
    float ocx, ocy, ofx, ofy;  // map_center_x/y, map_focal_x/y     (createMap.cl:5)
    float r[9];                // rot00..rot22                      (createMap.cl:6-8)
};

// atan for x >= 0 (or NaN): x > 1 -> pi/2 - atan(1/x); atan(t) = t + t*s*q(s), s = t*t.
// OpenCL leaves atan implementation defined; the algorithm is fixed here (and, independently,
// in the CPU oracle) so results are reproducible.  < 1.5 ulp.
__device__ __forceinline__ float atan_pos(float x) {
    const bool inv = x > 1.0f;
    const float t = inv ? 1.0f / x : x;
    const float s = t * t;
    float q = 0.0028423243202269077f;
    q = __builtin_fmaf(q, s, -0.016053270548582077f);
    q = __builtin_fmaf(q, s, 0.04269874095916748f);
    q = __builtin_fmaf(q, s, -0.07508683204650879f);
    q = __builtin_fmaf(q, s, 0.1064559817314148f);
    q = __builtin_fmaf(q, s, -0.14205896854400635f);
    q = __builtin_fmaf(q, s, 0.19993145763874054f);
    q = __builtin_fmaf(q, s, -0.33333125710487366f);
    float r = __builtin_fmaf(t * s, q, t);
    if (inv) r = (1.57079637050628662109375f - r) + (-4.37113900018624283e-8f);
    return r;
}

// Column / row terms of the rotated ray: dot(row_i, v) = (r_i0*vx + r_i1*vy) + r_i2*1
// (createMap.cl:27-31, left-to-right unfused).  r_i0*vx depends only on the column and
// r_i1*vy only on the row, so they are hoisted out of the per-pixel work.
struct ColTerm {
    float a0, a1, a2;
};
struct RowTerm {
    float b0, b1, b2;
};

__device__ __forceinline__ ColTerm col_term(const MapParams &p, int x) {
    const float vx = ((float)x - p.ocx) / p.ofx;  // createMap.cl:16
    return {p.r[0] * vx, p.r[3] * vx, p.r[6] * vx};
}
__device__ __forceinline__ RowTerm row_term(const MapParams &p, int y) {
    const float vy = ((float)y - p.ocy) / p.ofy;  // createMap.cl:17
    return {p.r[1] * vy, p.r[4] * vy, p.r[7] * vy};
}

__device__ __forceinline__ void map_pixel(const MapParams &p, const ColTerm &c, const RowTerm &r,
                                          float &mx, float &my) {
    const float wx = (c.a0 + r.b0) + p.r[2];
    const float wy = (c.a1 + r.b1) + p.r[5];
    const float wz = (c.a2 + r.b2) + p.r[8];
    const float px = wx / wz, py = wy / wz;             // createMap.cl:33-36
    const float rad = sqrtf(px * px + py * py);         // createMap.cl:38 length()
    const float k = atan_pos(rad) / rad;                // createMap.cl:39
    mx = p.icx + (px * k) * p.ifx;                      // createMap.cl:48
    my = p.icy + (py * k) * p.ify;                      // createMap.cl:49
}

// ---------------------------------------------------------------------------------------------
// cv::remap coordinate quantisation (OpenCV 4.5 CPU path): sx = cvRound(map*32),
// X = sx >> 5, f = sx & 31.  cvRound of NaN / out-of-int-range is INT_MIN on x86, which always
// lands outside the source; `far` reports those cases so the caller writes 0.
// ---------------------------------------------------------------------------------------------
struct Tap {
    int X, Y, fx, fy;
    bool far;
};

__device__ __forceinline__ Tap quantise(float mx, float my) {
    const float ax = mx * 32.0f, ay = my * 32.0f;
    Tap t;
    t.far = !(fabsf(ax) < 1073741824.0f) || !(fabsf(ay) < 1073741824.0f);
    const int sx = (int)__builtin_rintf(ax), sy = (int)__builtin_rintf(ay);
    t.X = sx >> 5, t.Y = sy >> 5, t.fx = sx & 31, t.fy = sy & 31;
    return t;
}

// ---------------------------------------------------------------------------------------------
// cvtColor(COLOR_YUV2BGR_NV12) arithmetic (OpenCV 4.5 CPU path): BT.601 limited
// range, 20-bit fixed point.
// ---------------------------------------------------------------------------------------------
constexpr int CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527;

struct ChromaTerm {
    int ruv, guv, buv;
};
__device__ __forceinline__ ChromaTerm chroma_term(int U, int V) {
    const int u = U - 128, v = V - 128;
    return {(1 << 19) + CVR * v, (1 << 19) + CVG * v + CUG * u, (1 << 19) + CUB * u};
}
__device__ __forceinline__ int sat8(int v) { return min(max(v, 0), 255); }

// ROCm 7.2 hipcc folds `sat8(a >> 20) | sat8(b >> 20) << 8` into gfx950's v_ashr_pk_u8_i32 and
// then ORs further bytes into the upper half of its result, which on MI355X is not zero (observed:
// 0xFFFF for negative inputs -> wrong bytes 2,3).  The empty asm makes the shifted value opaque so
// the clamp stays a plain v_med3_i32.
__device__ __forceinline__ int ashr20(int v) {
    v >>= 20;
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ void yuv_to_bgr(int Y, const ChromaTerm &c, int &b, int &g, int &r) {
    const int y = max(Y - 16, 0) * CY;
    b = sat8(ashr20(y + c.buv));
    g = sat8(ashr20(y + c.guv));
    r = sat8(ashr20(y + c.ruv));
}

}  // namespace vstab
