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
// Hand-scheduled IEEE-754 binary32 primitives for the hot kernel.  Each returns exactly what the
// correctly rounded operation returns for every finite, non-denormal-scaled input the map can
// produce (they are the LLVM AMDGPU expansions of fdiv / fsqrt minus the denormal pre-scaling;
// degenerate inputs -- 0, inf, NaN -- end in NaN or a huge value and the pixel is black either
// way).  tools/probe_isa.hip checks them against `/` and sqrtf on the GPU.
// ---------------------------------------------------------------------------------------------
// refined reciprocal: v_rcp_f32 (1 ulp) + one Newton step
__device__ __forceinline__ float rcp_refined(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
// n / d given r = rcp_refined(d): two residual corrections (Markstein), correctly rounded
__device__ __forceinline__ float div_with_rcp(float n, float d, float r) {
    float q = n * r;
    float e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r, q);
}
// correctly rounded sqrt: v_rsq_f32 seed, one coupled Newton step for sqrt and 1/(2 sqrt), one residual
// correction (LLVM's AMDGPU expansion of fsqrt for normal inputs; all plain FMAs).  x = 0 or inf give
// NaN here instead of 0 / inf -- both can only occur on a degenerate ray whose pixel is black either way.
__device__ __forceinline__ float sqrt_rn(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    const float e = __builtin_fmaf(-h, g, 0.5f);
    h = __builtin_fmaf(h, e, h);
    g = __builtin_fmaf(g, e, g);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}

// createMap.cl:13-50 with the primitives above; returns 32*map (exact power-of-two scaling folded
// into the constants: c32 = 32*src_center, f32 = 32*src_focal), i.e. the value cv::remap rounds.
struct MapParams32 {
    float icx32, icy32, ifx32, ify32;
    float r02, r12, r22;
};
__device__ __forceinline__ void map_pixel32(const MapParams32 &p, const ColTerm &c, const RowTerm &r,
                                            float &ax, float &ay) {
    const float wx = (c.a0 + r.b0) + p.r02;
    const float wy = (c.a1 + r.b1) + p.r12;
    const float wz = (c.a2 + r.b2) + p.r22;
    const float rz = rcp_refined(wz);
    const float px = div_with_rcp(wx, wz, rz), py = div_with_rcp(wy, wz, rz);
    const float rad = sqrt_rn(px * px + py * py);
    const float rr = rcp_refined(rad);
    // atan_pos(rad) with 1/rad from the shared refined reciprocal
    const bool inv = rad > 1.0f;
    const float t = inv ? div_with_rcp(1.0f, rad, rr) : rad;
    const float s = t * t;
    float q = 0.0028423243202269077f;
    q = __builtin_fmaf(q, s, -0.016053270548582077f);
    q = __builtin_fmaf(q, s, 0.04269874095916748f);
    q = __builtin_fmaf(q, s, -0.07508683204650879f);
    q = __builtin_fmaf(q, s, 0.1064559817314148f);
    q = __builtin_fmaf(q, s, -0.14205896854400635f);
    q = __builtin_fmaf(q, s, 0.19993145763874054f);
    q = __builtin_fmaf(q, s, -0.33333125710487366f);
    float at = __builtin_fmaf(t * s, q, t);
    at = inv ? (1.57079637050628662109375f - at) + (-4.37113900018624283e-8f) : at;
    const float k = div_with_rcp(at, rad, rr);
    ax = p.icx32 + (px * k) * p.ifx32;
    ay = p.icy32 + (py * k) * p.ify32;
}

// ---------------------------------------------------------------------------------------------
// The same arithmetic on two pixels at once (ext_vector_type(2) -> v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, which issue
// two fp32 operations in 4.8 cycles where two v_fma_f32 take 8; tools/probe_rate.hip).  Every component goes through exactly
// the operation sequence of the scalar functions above, so the results are bit-identical to them.
// ---------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return (f32x2){v, v}; }
__device__ __forceinline__ f32x2 rcp_refined2(f32x2 d) {
    const f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f32x2 e = fma2(-d, r, splat2(1.0f));
    return fma2(e, r, r);
}
__device__ __forceinline__ f32x2 div_with_rcp2(f32x2 n, f32x2 d, f32x2 r) {
    f32x2 q = n * r;
    f32x2 e = fma2(-d, q, n);
    q = fma2(e, r, q);
    e = fma2(-d, q, n);
    return fma2(e, r, q);
}
__device__ __forceinline__ f32x2 sqrt_rn2(f32x2 x) {
    const f32x2 y = {__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)};
    f32x2 g = x * y, h = splat2(0.5f) * y;
    const f32x2 e = fma2(-h, g, splat2(0.5f));
    h = fma2(h, e, h);
    g = fma2(g, e, g);
    const f32x2 d = fma2(-g, g, x);
    return fma2(d, h, g);
}
// map_pixel32 (FISH_TO_RECT = false) / map_pixel_ex<MAP_FISH_TO_RECT> (true) for two pixels of one column: a = column terms,
// b = row terms, r = third column of the rotation, each for the two rows (all three differ per row in the rolling-shutter warp)
template <bool FISH_TO_RECT>
__device__ __forceinline__ void map_pixel32_x2(float icx32, float icy32, float ifx32, float ify32, f32x2 r02, f32x2 r12, f32x2 r22,
                                               f32x2 a0, f32x2 a1, f32x2 a2, f32x2 b0, f32x2 b1, f32x2 b2, f32x2 &ax, f32x2 &ay) {
    const f32x2 wx = (a0 + b0) + r02;
    const f32x2 wy = (a1 + b1) + r12;
    const f32x2 wz = (a2 + b2) + r22;
    const f32x2 rz = rcp_refined2(wz);
    const f32x2 px = div_with_rcp2(wx, wz, rz), py = div_with_rcp2(wy, wz, rz);
    const f32x2 q = px * px + py * py;
    const f32x2 rad = sqrt_rn2(q);
    const f32x2 rr = rcp_refined2(rad);
    const i32x2 inv = rad > splat2(1.0f);
    const f32x2 t = inv ? div_with_rcp2(splat2(1.0f), rad, rr) : rad;
    const f32x2 s = t * t;
    f32x2 g = splat2(0.0028423243202269077f);
    g = fma2(g, s, splat2(-0.016053270548582077f));
    g = fma2(g, s, splat2(0.04269874095916748f));
    g = fma2(g, s, splat2(-0.07508683204650879f));
    g = fma2(g, s, splat2(0.1064559817314148f));
    g = fma2(g, s, splat2(-0.14205896854400635f));
    g = fma2(g, s, splat2(0.19993145763874054f));
    g = fma2(g, s, splat2(-0.33333125710487366f));
    f32x2 at = fma2(t * s, g, t);
    at = inv ? (splat2(1.57079637050628662109375f) - at) + splat2(-4.37113900018624283e-8f) : at;
    f32x2 k = div_with_rcp2(at, rad, rr);
    if constexpr (FISH_TO_RECT) k = (q == splat2(0.0f)) ? splat2(1.0f) : k;
    ax = splat2(icx32) + (px * k) * splat2(ifx32);
    ay = splat2(icy32) + (py * k) * splat2(ify32);
    if constexpr (FISH_TO_RECT) {
        const i32x2 ok = wz > splat2(0.0f);
        ax = ok ? ax : splat2(__builtin_nanf("")), ay = ok ? ay : splat2(__builtin_nanf(""));
    }
}

// ---------------------------------------------------------------------------------------------
// Generalised map (the libdewobble option surface in_p / out_p in {fish, rect},
// render.ts:611-617,669-683,711-717).  libdewobble is not part of the reference tree, so the arithmetic
// is defined by this project (the test suite holds a plain-C statement of it): mode
// FISH_TO_RECT performs createMap.cl's operations and differs only where createMap.cl degenerates
// (axis ray: correction factor 1 instead of 0/0; rays behind the camera: outside instead of mirrored).
// ---------------------------------------------------------------------------------------------
enum MapMode { MAP_CREATEMAP_CL = 0, MAP_FISH_TO_RECT = 1, MAP_FISH_TO_FISH = 2, MAP_RECT_TO_RECT = 3, MAP_RECT_TO_FISH = 4,
               // createMap.cl with the arithmetic ROCm's OpenCL compiler gives it on gfx950 (below)
               MAP_CREATEMAP_CL_OPENCL = 5,
               // internal: modes 0 / 1 / 5 with a per-row rotation (rolling shutter, BASELINE config 5)
               MAP_RS_CREATEMAP_CL = 6, MAP_RS_FISH_TO_RECT = 7, MAP_RS_CREATEMAP_CL_OPENCL = 8 };
// the projection pair (and arithmetic) of a mode: the rolling-shutter modes share their base mode's
constexpr bool map_mode_is_rs(int mode) { return mode >= MAP_RS_CREATEMAP_CL; }
constexpr int map_mode_base(int mode) {
    return mode == MAP_RS_CREATEMAP_CL ? MAP_CREATEMAP_CL : mode == MAP_RS_FISH_TO_RECT ? MAP_FISH_TO_RECT : mode == MAP_RS_CREATEMAP_CL_OPENCL ? MAP_CREATEMAP_CL_OPENCL : mode;
}
template <int MODE>
struct ModeTraits {
    static constexpr bool out_fish = MODE == MAP_FISH_TO_FISH || MODE == MAP_RECT_TO_FISH;
    static constexpr bool in_fish = MODE == MAP_FISH_TO_RECT || MODE == MAP_FISH_TO_FISH || MODE == MAP_CREATEMAP_CL || MODE == MAP_CREATEMAP_CL_OPENCL;
};

// ---------------------------------------------------------------------------------------------
// createMap.cl:13-50 AS THE REFERENCE'S OWN KERNEL COMPUTES IT ON THIS GPU.  The reference hands createMap.cl to the
// OpenCL runtime (FrameSourceWarp.cpp:224,301), whose compiler is free to contract a*b+c (FP_CONTRACT is ON in OpenCL
// C), to divide within 2.5 ulp, to take length() within 3 and atan() within 5.  The test infrastructure holds
// that file built by ROCm's OpenCL front end for gfx950 (createMap.gfx950.co); its instruction stream is (llvm-objdump -d):
//   a / b      = ldexp(frexp_mant(a) * v_rcp_f32(frexp_mant(b)), frexp_exp(a) - frexp_exp(b))
//   dot(r, v)  = fma(r1, vy, r0 * vx) + r2           (the * 1 of the third component folded away)
//   length(p)  = v_sqrt_f32(fma(py, py, px * px))      with ocml's rescaling outside [2^-126, inf)
//   atan(x)    = ocml's: t = x > 1 ? v_rcp_f32(x) : x, odd polynomial of degree 17 in t, pi/2 - . when reciprocated
//   map        = fma(focal, p * k, centre)
// The functions below are that stream, operation for operation; tests/test_refcl_gpu.py compares them bit for bit with
// the code object run on the same device.  v_rcp_f32 / v_sqrt_f32 are hardware approximations (1 ulp), so this mode
// has no CPU restatement: its checker is the reference's own kernel.
// ---------------------------------------------------------------------------------------------
constexpr float f32_bits(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ float ocl_div(float a, float b) {
    const float r = __builtin_amdgcn_rcpf(__builtin_amdgcn_frexp_mantf(b));
    const float m = __builtin_amdgcn_frexp_mantf(a) * r;
    return __builtin_ldexpf(m, __builtin_amdgcn_frexp_expf(a) - __builtin_amdgcn_frexp_expf(b));
}
__device__ __forceinline__ float ocl_sqrt_scaled(float x) {  // ocml's native-precision sqrt: denormal inputs rescaled
    const bool small = 0x1p-126f > x;
    const float y = __builtin_amdgcn_sqrtf(__builtin_ldexpf(x, small ? 32 : 0));
    return __builtin_ldexpf(y, small ? -16 : 0);
}
__device__ __forceinline__ float ocl_length2(float px, float py) {
    const float q = __builtin_fmaf(py, py, px * px);
    if (!(0x1p-126f > q)) {  // also NaN
        if (q != __builtin_inff()) return __builtin_amdgcn_sqrtf(q);
        const float a = px * 0x1p-65f, b = py * 0x1p-65f;
        return 0x1p65f * ocl_sqrt_scaled(__builtin_fmaf(b, b, a * a));
    }
    const float a = px * 0x1p86f, b = py * 0x1p86f;
    return 0x1p-86f * ocl_sqrt_scaled(__builtin_fmaf(b, b, a * a));
}
// ocml atan, given t = |x| > 1 ? 1 / |x| : |x| (the reciprocal is a raw v_rcp_f32 there)
__device__ __forceinline__ float ocl_atan_poly(float t, bool inv) {
    const float s = t * t;
    float p = __builtin_fmaf(f32_bits(0x3b2d2a58u), s, f32_bits(0xbc7a590cu));
    p = __builtin_fmaf(s, p, f32_bits(0x3d29fb3fu));
    p = __builtin_fmaf(s, p, f32_bits(0xbd97d4d7u));
    p = __builtin_fmaf(s, p, f32_bits(0x3dd931b2u));
    p = __builtin_fmaf(s, p, f32_bits(0xbe1160e6u));
    p = __builtin_fmaf(s, p, f32_bits(0x3e4cb8bfu));
    p = __builtin_fmaf(s, p, f32_bits(0xbeaaaa62u));
    const float r = __builtin_fmaf(t, s * p, t);
    return inv ? f32_bits(0x3fc90fdbu) - r : r;
}
__device__ __forceinline__ float ocl_atan(float x) {
    const float a = __builtin_fabsf(x);
    const bool inv = a > 1.0f;
    return __builtin_copysignf(ocl_atan_poly(inv ? __builtin_amdgcn_rcpf(a) : a, inv), x);
}
// One pixel, the code object's stream literally.  vx, vy = ocl_div(x - ocx, ofx), ocl_div(y - ocy, ofy); a = r_i0 * vx.
// c = centre, f = focal length of the input camera (both may carry cv::remap's exact factor 32).
// r = the nine rotation entries (the frame's, or one output row's in the rolling-shutter warp).
__device__ __forceinline__ void map_pixel_ocl_literal(float icx, float icy, float ifx, float ify, const float (&r)[9], float a0, float a1, float a2,
                                                      float vy, float &ax, float &ay) {
    const float wz = __builtin_fmaf(r[7], vy, a2) + r[8];
    const float wx = __builtin_fmaf(r[1], vy, a0) + r[2];
    const float wy = __builtin_fmaf(r[4], vy, a1) + r[5];
    const float px = ocl_div(wx, wz), py = ocl_div(wy, wz);
    const float rad = ocl_length2(px, py);
    const float k = ocl_div(ocl_atan(rad), rad);
    ax = __builtin_fmaf(ifx, px * k, icx);
    ay = __builtin_fmaf(ify, py * k, icy);
}
// The same results from fewer instructions wherever no intermediate leaves the normal range: a multiplication's
// rounding does not depend on its operands' exponents and v_rcp_f32 works on the significand alone, so
// frexp / ldexp around `a * rcp(b)` change nothing, and the reciprocal of the radius serves both atan's argument
// reduction and the final division.  `regular` says whether that holds for this pixel (else: the literal stream).
__device__ __forceinline__ bool map_pixel_ocl_fast(float icx, float icy, float ifx, float ify, const float (&r)[9], float a0, float a1, float a2,
                                                   float vy, float &ax, float &ay) {
    const float wz = __builtin_fmaf(r[7], vy, a2) + r[8];
    const float wx = __builtin_fmaf(r[1], vy, a0) + r[2];
    const float wy = __builtin_fmaf(r[4], vy, a1) + r[5];
    const float rz = __builtin_amdgcn_rcpf(wz);
    const float px = wx * rz, py = wy * rz;
    const float q = __builtin_fmaf(py, py, px * px);
    const float rad = __builtin_amdgcn_sqrtf(q), rr = __builtin_amdgcn_rcpf(rad);
    const bool inv = rad > 1.0f;
    const float k = ocl_atan_poly(inv ? rr : rad, inv) * rr;
    ax = __builtin_fmaf(ifx, px * k, icx);
    ay = __builtin_fmaf(ify, py * k, icy);
    const float az = __builtin_fabsf(wz);
    return az >= 0x1p-40f && az <= 0x1p40f && q >= 0x1p-80f && q <= 0x1p80f;  // false for NaN
}

// sin, cos on [0, pi]: quadrant reduction with a two-constant pi/2, Cephes single-precision polynomials
__device__ __forceinline__ void sincos_pos(float t, float &sn, float &cs) {
    const float k = __builtin_rintf(t * 0.636619746685028076171875f);
    float r = __builtin_fmaf(k, -1.57079637050628662109375f, t);
    r = __builtin_fmaf(k, 4.37113900018624283e-8f, r);
    const float z = r * r;
    const float ps = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    const float pc = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    const float s = __builtin_fmaf(r * z, ps, r);
    const float c = __builtin_fmaf(z * z, pc, __builtin_fmaf(-0.5f, z, 1.0f));
    sn = k == 1.0f ? c : k == 2.0f ? -s : s;
    cs = k == 1.0f ? -s : k == 2.0f ? -c : c;
}

// Normalised output coordinate (createMap.cl:16-17) as the mode divides: IEEE, or the OpenCL build's reciprocal form.
template <int MODE>
__device__ __forceinline__ float norm_coord(float n, float d, float rcp_refined_d) {
    if constexpr (MODE == MAP_CREATEMAP_CL_OPENCL) return ocl_div(n, d);
    else return div_with_rcp(n, d, rcp_refined_d);
}

// One pixel of the generalised map.  p holds the input camera (scaled by 32 in the fused kernel, unscaled in
// the map-plane kernel -- the scaling is an exact power of two either way), P the output camera and rotation.
// c / r are the hoisted column / row products for pinhole output; vx, vy the normalised output coordinates for
// fisheye output.  Outside (NaN) when the output ray is beyond 180 degrees or lands behind the input camera.
template <int MODE>
__device__ __forceinline__ void map_pixel_ex(const MapParams32 &p, const MapParams &P, const ColTerm &c, const RowTerm &r,
                                             float vx, float vy, float &ax, float &ay) {
    if constexpr (MODE == MAP_CREATEMAP_CL) {
        map_pixel32(p, c, r, ax, ay);
    } else if constexpr (MODE == MAP_CREATEMAP_CL_OPENCL) {
        if (!map_pixel_ocl_fast(p.icx32, p.icy32, p.ifx32, p.ify32, P.r, c.a0, c.a1, c.a2, vy, ax, ay))
            map_pixel_ocl_literal(p.icx32, p.icy32, p.ifx32, p.ify32, P.r, c.a0, c.a1, c.a2, vy, ax, ay);
    } else {
        float wx, wy, wz;
        bool ok = true;
        if constexpr (ModeTraits<MODE>::out_fish) {
            const float q = vx * vx + vy * vy;
            const bool zero = q == 0.0f;
            const float rho = sqrt_rn(q);  // NaN when q == 0
            ok = zero || rho < 3.1415927410125732421875f;
            float sn, cs;
            sincos_pos(zero ? 0.0f : rho, sn, cs);
            const float s = zero ? 1.0f : div_with_rcp(sn, rho, rcp_refined(rho));
            const float rx = vx * s, ry = vy * s;
            wx = (P.r[0] * rx + P.r[1] * ry) + P.r[2] * cs;
            wy = (P.r[3] * rx + P.r[4] * ry) + P.r[5] * cs;
            wz = (P.r[6] * rx + P.r[7] * ry) + P.r[8] * cs;
        } else {
            wx = (c.a0 + r.b0) + p.r02;
            wy = (c.a1 + r.b1) + p.r12;
            wz = (c.a2 + r.b2) + p.r22;
        }
        ok = ok && wz > 0.0f;
        const float rz = rcp_refined(wz);
        const float px = div_with_rcp(wx, wz, rz), py = div_with_rcp(wy, wz, rz);
        if constexpr (ModeTraits<MODE>::in_fish) {
            const float q = px * px + py * py;
            const bool zero = q == 0.0f;
            const float rad = sqrt_rn(q);
            const float rr = rcp_refined(rad);
            const bool inv = rad > 1.0f;
            const float t = inv ? div_with_rcp(1.0f, rad, rr) : rad;
            const float s = t * t;
            float g = 0.0028423243202269077f;
            g = __builtin_fmaf(g, s, -0.016053270548582077f);
            g = __builtin_fmaf(g, s, 0.04269874095916748f);
            g = __builtin_fmaf(g, s, -0.07508683204650879f);
            g = __builtin_fmaf(g, s, 0.1064559817314148f);
            g = __builtin_fmaf(g, s, -0.14205896854400635f);
            g = __builtin_fmaf(g, s, 0.19993145763874054f);
            g = __builtin_fmaf(g, s, -0.33333125710487366f);
            float at = __builtin_fmaf(t * s, g, t);
            at = inv ? (1.57079637050628662109375f - at) + (-4.37113900018624283e-8f) : at;
            const float k = zero ? 1.0f : div_with_rcp(at, rad, rr);
            ax = p.icx32 + (px * k) * p.ifx32;
            ay = p.icy32 + (py * k) * p.ify32;
        } else {
            ax = p.icx32 + px * p.ifx32;
            ay = p.icy32 + py * p.ify32;
        }
        if (!ok) ax = ay = __builtin_nanf("");
    }
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
// OpenCV 4.5 cvtColor(COLOR_BGR2YUV_I420) arithmetic (RGB8toYUV420pInvoker): BT.601 limited range, 20-bit
// fixed point; chroma from the top-left pixel of each 2x2 block.  No saturation is needed: Y in [16, 235],
// U / V in [16, 240] for every 8-bit BGR.  v = B | G << 8 | R << 16.
constexpr int CRY = 269484, CGY = 528482, CBY = 102760, CRU = -155188, CGU = -305135, CBU = 460324, CGV = -385875, CBV = -74448;
__device__ __forceinline__ uint32_t bgr_to_y(uint32_t v) {
    const int B = v & 255, G = (v >> 8) & 255, R = (v >> 16) & 255;
    return (uint32_t)(CRY * R + CGY * G + CBY * B + (1 << 19) + (16 << 20)) >> 20;
}
__device__ __forceinline__ uint32_t bgr_to_uv(uint32_t v) {  // U | V << 8
    const int B = v & 255, G = (v >> 8) & 255, R = (v >> 16) & 255;
    const uint32_t U = (uint32_t)(CRU * R + CGU * G + CBU * B + (1 << 19) + (128 << 20)) >> 20;
    const uint32_t V = (uint32_t)(CBU * R + CGV * G + CBV * B + (1 << 19) + (128 << 20)) >> 20;
    return U | (V << 8);
}
__device__ __forceinline__ int sat8(int v) { return min(max(v, 0), 255); }

// gfx950's v_ashr_pk_u8_i32 D, S0, S1, n writes {sat_u8(S0 >> n), sat_u8(S1 >> n)} into ONE 16-bit
// half of D (low half; high half with op_sel:[0,0,0,1]) and PRESERVES the other half (measured:
// tools/probe_isa.hip).  ROCm 7.2 hipcc pattern-matches `sat8(a >> 20) | sat8(b >> 20) << 8` into
// it but then assumes the other half is zero -> wrong bytes 2,3.  The empty asm below makes the
// shifted value opaque so the generic path keeps a plain v_med3_i32; the tiled kernel uses the
// instruction deliberately (pack_bgrx).
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


// Chroma terms with the luma offset folded in: channel = sat8((max(Y,16)*CY + term) >> 20), which is
// the same integer as sat8((max(Y-16,0)*CY + (1<<19) + C*uv) >> 20).
__device__ __forceinline__ ChromaTerm chroma_term_folded(int U, int V) {
    const int u = U - 128, v = V - 128;
    constexpr int K = (1 << 19) - 16 * CY;
    return {K + CVR * v, K + CVG * v + CUG * u, K + CUB * u};
}

// One BGRx pixel (byte 3 = 0) from a luma byte and folded chroma terms: 1 max + 3 mad + 2 pack.
__device__ __forceinline__ uint32_t pack_bgrx(int Y, const ChromaTerm &c) {
    const int y = max(Y, 16);
    const int b = __mul24(y, CY) + c.buv, g = __mul24(y, CY) + c.guv, r = __mul24(y, CY) + c.ruv;
    uint32_t d;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 20" : "=v"(d) : "v"(b), "v"(g));
    asm("v_ashr_pk_u8_i32 %0, %1, 0, 20 op_sel:[0,0,0,1]" : "+v"(d) : "v"(r));
    return d;
}

// cv::remap fixed-point bilinear blend of four BGRx taps:
//   out_c = (p00*(32-fx)(32-fy) + p01*fx(32-fy) + p10*(32-fx)fy + p11*fx*fy + 512) >> 10
// evaluated as an exact two-stage integer lerp (vertical per column with v_dot4_u32_u8 on byte pairs gathered by
// v_perm_b32, then horizontal per channel).  Integer arithmetic is distributive, so this equals the four-product
// sum bit for bit.  Returns 0x00RRGGBB.
__device__ __forceinline__ uint32_t blend_bgrx(uint32_t t00, uint32_t t01, uint32_t t10, uint32_t t11,
                                               uint32_t fx, uint32_t fy) {
    // vertical lerp per column and channel as byte dot products: v = top * (32 - fy) + bottom * fy  (<= 255 * 32)
    const uint32_t gy = 32u - fy;
    const uint32_t w_lo = gy | (fy << 8), w_hi = w_lo << 16;  // weights against bytes (0,1) / bytes (2,3)
    const uint32_t bg_l = __builtin_amdgcn_perm(t10, t00, 0x05010400u);  // [top.B, bottom.B, top.G, bottom.G]
    const uint32_t r_l = __builtin_amdgcn_perm(t10, t00, 0x0c0c0602u);   // [top.R, bottom.R, 0, 0]
    const uint32_t bg_r = __builtin_amdgcn_perm(t11, t01, 0x05010400u);
    const uint32_t r_r = __builtin_amdgcn_perm(t11, t01, 0x0c0c0602u);
    // horizontal lerp with the weights scaled by 64: (l * gx + r * fx + 512) << 6 has the result byte at bits 16..23.
    // The six dot products and six multiply-adds are ONE hand-ordered block: gfx950 needs three independent
    // instructions between a v_dot4 and the VALU instruction that reads its result (the compiler pads its own dot4s with
    // s_nop, cannot see into inline assembly, and left to itself emits two multiplies and a three-operand add per channel
    // where two multiply-adds do) -- here every dot product is read five instructions after it was issued.
    const uint32_t fx6 = fx << 6, gx6 = 2048u - fx6, half = 32768u;
    uint32_t vb_r, vg_r, vr_r, vb_l, vg_l, vr_l, sb, sg, sr;
    asm("v_dot4_u32_u8 %0, %9, %13, 0\n\t"
        "v_dot4_u32_u8 %1, %9, %14, 0\n\t"
        "v_dot4_u32_u8 %2, %10, %13, 0\n\t"
        "v_dot4_u32_u8 %3, %11, %13, 0\n\t"
        "v_dot4_u32_u8 %4, %11, %14, 0\n\t"
        "v_dot4_u32_u8 %5, %12, %13, 0\n\t"
        "v_mad_u32_u24 %6, %0, %15, %17\n\t"
        "v_mad_u32_u24 %7, %1, %15, %17\n\t"
        "v_mad_u32_u24 %8, %2, %15, %17\n\t"
        "v_mad_u32_u24 %6, %3, %16, %6\n\t"
        "v_mad_u32_u24 %7, %4, %16, %7\n\t"
        "v_mad_u32_u24 %8, %5, %16, %8"
        : "=&v"(vb_r), "=&v"(vg_r), "=&v"(vr_r), "=&v"(vb_l), "=&v"(vg_l), "=&v"(vr_l), "=&v"(sb), "=&v"(sg), "=&v"(sr)
        : "v"(bg_r), "v"(r_r), "v"(bg_l), "v"(r_l), "v"(w_lo), "v"(w_hi), "v"(fx6), "v"(gx6), "v"(half));
    const uint32_t bg = __builtin_amdgcn_perm(sg, sb, 0x0c0c0602u);  // [B, G, 0, 0]
    return __builtin_amdgcn_perm(sr, bg, 0x0c060100u);               // [B, G, R, 0]
}

}  // namespace vstab
