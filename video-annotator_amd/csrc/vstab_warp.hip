// vstab_warp.hip -- pixel-path HIP kernels for gfx950 (MI355X): NV12 packing, NV12->BGR,
// createMap, remap, and the fused undistort-remap kernel.  Compiled with -ffp-contract=off.
//
// All of this is HBM-bound byte/gather work (no MFMA): the design rules that matter are
// coalesced wide accesses, enough workgroups to fill 256 CUs, and keeping the map in registers.
#include <climits>
#include <cstdlib>

#include <hip/hip_ext.h>

#include "vstab_device.hpp"
#include "vstab_internal.hpp"
#include "vstab_warp_args.hpp"

namespace vstab {

// =============================================================================================
// k_pack_nv12 -- replaces the two clEnqueueCopyImageToBuffer calls of
// FrameSourceFfmpegOpenCl.cpp:64-85.  One launch copies both planes: rows [0,h) come from the
// luma plane, rows [h, h*3/2) from the chroma plane.  16 B per lane when every pitch and base
// is 16-B aligned, bytes otherwise.
// =============================================================================================
template <typename V>
__global__ void __launch_bounds__(256) k_pack_nv12(const uint8_t *__restrict__ y, size_t pitch_y,
                                                   const uint8_t *__restrict__ uv, size_t pitch_uv,
                                                   int row_vecs, int h, uint8_t *__restrict__ dst,
                                                   size_t pitch_dst) {
    // flat grid-stride loop over (row, vector) pairs: a small persistent grid keeps the copy from
    // taking wave slots away from kernels running beside it on other streams
    const int rows = h + h / 2;
    const long total = (long)rows * row_vecs;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int row = (int)(e / row_vecs), i = (int)(e - (long)row * row_vecs);
        const uint8_t *s = row < h ? y + (size_t)row * pitch_y : uv + (size_t)(row - h) * pitch_uv;
        uint8_t *d = dst + (size_t)row * pitch_dst;
        reinterpret_cast<V *>(d)[i] = reinterpret_cast<const V *>(s)[i];
    }
}

// =============================================================================================
// k_pack_p010 -- 10-bit input (BASELINE config 5: P010 / P016 planes, 16-bit little-endian samples with the
// significant bits at the top) narrowed to the 8-bit packed NV12 the reference path works on: byte = sample >> 8.
// No reference counterpart (the reference only ever sees 8-bit NV12, FrameSourceFfmpegOpenCl.cpp:53-56).
// One thread narrows 8 samples (one 16-B load, one 8-B store) when everything is aligned, else one sample.
// =============================================================================================
template <bool VEC>
__global__ void __launch_bounds__(256) k_pack_p010(const uint8_t *__restrict__ y, size_t pitch_y,
                                                   const uint8_t *__restrict__ uv, size_t pitch_uv, int row_units,
                                                   int h, int rows, uint8_t *__restrict__ dst, size_t pitch_dst) {
    const long total = (long)rows * row_units;  // rows = h + h / 2, or h when only the luma plane is wanted
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int row = (int)(e / row_units), i = (int)(e - (long)row * row_units);
        const uint8_t *s = row < h ? y + (size_t)row * pitch_y : uv + (size_t)(row - h) * pitch_uv;
        uint8_t *d = dst + (size_t)row * pitch_dst;
        if constexpr (VEC) {
            const uint4 v = reinterpret_cast<const uint4 *>(s)[i];
            const uint32_t lo = __builtin_amdgcn_perm(v.y, v.x, 0x07050301u), hi = __builtin_amdgcn_perm(v.w, v.z, 0x07050301u);
            reinterpret_cast<uint2 *>(d)[i] = make_uint2(lo, hi);
        } else {
            d[i] = s[2 * i + 1];  // high byte of the little-endian sample
        }
    }
}

// =============================================================================================
// k_cvt_nv12_bgr -- cvtColor(COLOR_YUV2BGR_NV12), FrameSourceWarp.cpp:401.  Compatibility /
// parity kernel (the fused path never materialises the BGR frame).  One thread converts an
// 8 x 2 luma block: two 8-B luma loads, one 8-B chroma load, two 24-B BGR stores.
// =============================================================================================
__global__ void __launch_bounds__(256) k_cvt_nv12_bgr(const uint8_t *__restrict__ yp, size_t pitch_y,
                                                      const uint8_t *__restrict__ uvp, size_t pitch_uv,
                                                      int w, int h, uint8_t *__restrict__ dst,
                                                      size_t pitch_dst, int vec_ok) {
    const int bx = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const int by = (blockIdx.y * blockDim.y + threadIdx.y) * 2;
    if (bx >= w || by >= h) return;
    const uint8_t *y0 = yp + (size_t)by * pitch_y + bx;
    const uint8_t *y1 = y0 + pitch_y;
    const uint8_t *uv = uvp + (size_t)(by >> 1) * pitch_uv + bx;
    uint8_t *d0 = dst + (size_t)by * pitch_dst + (size_t)bx * 3;
    uint8_t *d1 = d0 + pitch_dst;
    if (vec_ok && bx + 8 <= w) {
        const uint2 a = *reinterpret_cast<const uint2 *>(y0);
        const uint2 b = *reinterpret_cast<const uint2 *>(y1);
        const uint2 c = *reinterpret_cast<const uint2 *>(uv);
        const uint32_t ya[2] = {a.x, a.y}, yb[2] = {b.x, b.y}, cc[2] = {c.x, c.y};
        uint32_t o0[6], o1[6];
        uint8_t t0[24], t1[24];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int pair = i >> 1;
            const uint32_t cw = cc[pair >> 1] >> ((pair & 1) * 16);
            const ChromaTerm ct = chroma_term(cw & 255, (cw >> 8) & 255);
            int bb, gg, rr;
            yuv_to_bgr((ya[i >> 2] >> ((i & 3) * 8)) & 255, ct, bb, gg, rr);
            t0[3 * i] = bb, t0[3 * i + 1] = gg, t0[3 * i + 2] = rr;
            yuv_to_bgr((yb[i >> 2] >> ((i & 3) * 8)) & 255, ct, bb, gg, rr);
            t1[3 * i] = bb, t1[3 * i + 1] = gg, t1[3 * i + 2] = rr;
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {
            o0[i] = t0[4 * i] | (t0[4 * i + 1] << 8) | (t0[4 * i + 2] << 16) | ((uint32_t)t0[4 * i + 3] << 24);
            o1[i] = t1[4 * i] | (t1[4 * i + 1] << 8) | (t1[4 * i + 2] << 16) | ((uint32_t)t1[4 * i + 3] << 24);
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            reinterpret_cast<uint2 *>(d0)[i] = make_uint2(o0[2 * i], o0[2 * i + 1]);
            reinterpret_cast<uint2 *>(d1)[i] = make_uint2(o1[2 * i], o1[2 * i + 1]);
        }
    } else {
        for (int i = 0; i < 8 && bx + i < w; i++) {
            const ChromaTerm ct = chroma_term(uv[i & ~1], uv[(i & ~1) + 1]);
            int bb, gg, rr;
            yuv_to_bgr(y0[i], ct, bb, gg, rr);
            d0[3 * i] = bb, d0[3 * i + 1] = gg, d0[3 * i + 2] = rr;
            if (by + 1 < h) {
                yuv_to_bgr(y1[i], ct, bb, gg, rr);
                d1[3 * i] = bb, d1[3 * i + 1] = gg, d1[3 * i + 2] = rr;
            }
        }
    }
}

// =============================================================================================
// k_create_map -- createMap.cl:1-51 as a stand-alone kernel (parity / un-fused mode only).
// 16x16 threads, 4 columns per thread (one 16-B store per plane when aligned).
// =============================================================================================
__global__ void __launch_bounds__(256) k_create_map(float *__restrict__ mapx, size_t pitch_x,
                                                    float *__restrict__ mapy, size_t pitch_y,
                                                    int cols, int rows, MapParams p, int vec_ok) {
    const int x0 = (blockIdx.x * 16 + threadIdx.x) * 4;
    const int y = blockIdx.y * 16 + threadIdx.y;
    if (x0 >= cols || y >= rows) return;
    const RowTerm rt = row_term(p, y);
    float mx[4], my[4];
#pragma unroll
    for (int i = 0; i < 4; i++) map_pixel(p, col_term(p, x0 + i), rt, mx[i], my[i]);
    float *px = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(mapx) + (size_t)y * pitch_x) + x0;
    float *py = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(mapy) + (size_t)y * pitch_y) + x0;
    if (vec_ok && x0 + 4 <= cols) {
        *reinterpret_cast<float4 *>(px) = make_float4(mx[0], mx[1], mx[2], mx[3]);
        *reinterpret_cast<float4 *>(py) = make_float4(my[0], my[1], my[2], my[3]);
    } else {
        for (int i = 0; i < 4 && x0 + i < cols; i++) px[i] = mx[i], py[i] = my[i];
    }
}

// k_create_map_ex -- the generalised map (modes 1..4, vstab_device.hpp map_pixel_ex) as map planes.
template <int MODE>
__global__ void __launch_bounds__(256) k_create_map_ex(float *__restrict__ mapx, size_t pitch_x,
                                                       float *__restrict__ mapy, size_t pitch_y,
                                                       int cols, int rows, MapParams p, int vec_ok) {
    const int x0 = (blockIdx.x * 16 + threadIdx.x) * 4;
    const int y = blockIdx.y * 16 + threadIdx.y;
    if (x0 >= cols || y >= rows) return;
    const MapParams32 in = {p.icx, p.icy, p.ifx, p.ify, p.r[2], p.r[5], p.r[8]};  // unscaled here
    constexpr bool OCL = MODE == MAP_CREATEMAP_CL_OPENCL;
    const float vy = OCL ? ocl_div((float)y - p.ocy, p.ofy) : ((float)y - p.ocy) / p.ofy;
    const RowTerm rt = {p.r[1] * vy, p.r[4] * vy, p.r[7] * vy};
    float mx[4], my[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float vx = OCL ? ocl_div((float)(x0 + i) - p.ocx, p.ofx) : ((float)(x0 + i) - p.ocx) / p.ofx;
        const ColTerm ct = {p.r[0] * vx, p.r[3] * vx, p.r[6] * vx};
        // the map-plane operator runs the OpenCL build's instruction stream literally; the fused kernel and the
        // quantised map use its shortened form with this one as the fall-back (vstab_device.hpp)
        if constexpr (OCL) map_pixel_ocl_literal(p.icx, p.icy, p.ifx, p.ify, p.r, ct.a0, ct.a1, ct.a2, vy, mx[i], my[i]);
        else map_pixel_ex<MODE>(in, p, ct, rt, vx, vy, mx[i], my[i]);
    }
    float *px = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(mapx) + (size_t)y * pitch_x) + x0;
    float *py = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(mapy) + (size_t)y * pitch_y) + x0;
    if (vec_ok && x0 + 4 <= cols) {
        *reinterpret_cast<float4 *>(px) = make_float4(mx[0], mx[1], mx[2], mx[3]);
        *reinterpret_cast<float4 *>(py) = make_float4(my[0], my[1], my[2], my[3]);
    } else {
        for (int i = 0; i < 4 && x0 + i < cols; i++) px[i] = mx[i], py[i] = my[i];
    }
}

// =============================================================================================
// k_remap_bilinear -- cv::remap(INTER_LINEAR, BORDER_CONSTANT 0), FrameSourceWarp.cpp:306-312,
// reading float map planes (un-fused compatibility mode).  One thread per output pixel.
// =============================================================================================
template <int CN>
__global__ void __launch_bounds__(256) k_remap_bilinear(const uint8_t *__restrict__ src, size_t pitch_src,
                                                        int sw, int sh, const float *__restrict__ mapx,
                                                        size_t pitch_x, const float *__restrict__ mapy,
                                                        size_t pitch_y, uint8_t *__restrict__ dst,
                                                        size_t pitch_dst, int dw, int dh) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) return;
    const float mx = reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(mapx) + (size_t)y * pitch_x)[x];
    const float my = reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(mapy) + (size_t)y * pitch_y)[x];
    const Tap t = quantise(mx, my);
    uint8_t *o = dst + (size_t)y * pitch_dst + (size_t)x * CN;
    if (t.far || t.X >= sw || t.X + 1 < 0 || t.Y >= sh || t.Y + 1 < 0) {
#pragma unroll
        for (int c = 0; c < CN; c++) o[c] = 0;
        return;
    }
    const int w00 = (32 - t.fx) * (32 - t.fy), w01 = t.fx * (32 - t.fy), w10 = (32 - t.fx) * t.fy,
              w11 = t.fx * t.fy;
    const bool x0 = t.X >= 0, x1 = t.X + 1 < sw, y0 = t.Y >= 0, y1 = t.Y + 1 < sh;
    const uint8_t *r0 = src + (size_t)max(t.Y, 0) * pitch_src, *r1 = src + (size_t)min(t.Y + 1, sh - 1) * pitch_src;
    const size_t c0 = (size_t)max(t.X, 0) * CN, c1 = (size_t)min(t.X + 1, sw - 1) * CN;
#pragma unroll
    for (int c = 0; c < CN; c++) {
        const int p00 = (x0 && y0) ? r0[c0 + c] : 0, p01 = (x1 && y0) ? r0[c1 + c] : 0;
        const int p10 = (x0 && y1) ? r1[c0 + c] : 0, p11 = (x1 && y1) ? r1[c1 + c] : 0;
        o[c] = (uint8_t)((p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + 512) >> 10);
    }
}

// =============================================================================================
// k_warp_nv12_bgr (v1, direct gather) -- the fused hot kernel:
//   createMap.cl:13-50  ->  remap quantisation (:306-312)  ->  4 NV12 taps converted with the
//   cvtColor arithmetic (:401)  ->  fixed-point bilinear blend  ->  BGR8.
// Converting each tap and then blending is exactly what the reference's cvtColor-then-remap
// sequence computes, so the output is bit-identical to the un-fused operators.
//
// Block = 16 x 16 threads; a thread owns 4 adjacent columns (12 output bytes = 3 dwords per
// row) and ROWS_PER_THREAD rows 16 apart, so the column terms of the rotated ray (and their
// divisions) are computed once per thread and the row terms once per row.
// =============================================================================================
constexpr int WARP_ROWS_PER_THREAD = 4;
constexpr int WARP_TILE_W = 64, WARP_TILE_H = 16 * WARP_ROWS_PER_THREAD;

__device__ __forceinline__ uint32_t warp_pixel(const WarpArgs &a, const ColTerm &ct, const RowTerm &rt) {
    float mx, my;
    map_pixel(a.p, ct, rt, mx, my);
    const Tap t = quantise(mx, my);
    if (t.far || t.X >= a.sw || t.X + 1 < 0 || t.Y >= a.sh || t.Y + 1 < 0) return 0;
    const int w00 = (32 - t.fx) * (32 - t.fy), w01 = t.fx * (32 - t.fy), w10 = (32 - t.fx) * t.fy,
              w11 = t.fx * t.fy;
    int b0, g0, r0, b1, g1, r1, b2, g2, r2, b3, g3, r3;
    fetch_tap(a, t.X, t.Y, b0, g0, r0);
    fetch_tap(a, t.X + 1, t.Y, b1, g1, r1);
    fetch_tap(a, t.X, t.Y + 1, b2, g2, r2);
    fetch_tap(a, t.X + 1, t.Y + 1, b3, g3, r3);
    const uint32_t B = (uint32_t)(b0 * w00 + b1 * w01 + b2 * w10 + b3 * w11 + 512) >> 10;
    const uint32_t G = (uint32_t)(g0 * w00 + g1 * w01 + g2 * w10 + g3 * w11 + 512) >> 10;
    const uint32_t R = (uint32_t)(r0 * w00 + r1 * w01 + r2 * w10 + r3 * w11 + 512) >> 10;
    return B | (G << 8) | (R << 16);
}

__global__ void __launch_bounds__(256) k_warp_nv12_bgr(WarpArgs a, int vec_ok) {
    const int x0 = blockIdx.x * WARP_TILE_W + threadIdx.x * 4;
    if (x0 >= a.dw) return;
    ColTerm ct[4];
#pragma unroll
    for (int i = 0; i < 4; i++) ct[i] = col_term(a.p, x0 + i);
#pragma unroll 1
    for (int j = 0; j < WARP_ROWS_PER_THREAD; j++) {
        const int y = blockIdx.y * WARP_TILE_H + j * 16 + threadIdx.y;
        if (y >= a.dh) break;
        const RowTerm rt = row_term(a.p, y);
        uint32_t px[4];
#pragma unroll
        for (int i = 0; i < 4; i++) px[i] = warp_pixel(a, ct[i], rt);
        uint8_t *o = a.dst + (size_t)y * a.pitch_dst + (size_t)x0 * 3;
        if (vec_ok && x0 + 4 <= a.dw) {
            uint32_t *o32 = reinterpret_cast<uint32_t *>(o);
            o32[0] = px[0] | (px[1] << 24);
            o32[1] = (px[1] >> 8) | (px[2] << 16);
            o32[2] = (px[2] >> 16) | (px[3] << 8);
        } else {
            for (int i = 0; i < 4 && x0 + i < a.dw; i++) {
                o[3 * i] = px[i] & 255, o[3 * i + 1] = (px[i] >> 8) & 255, o[3 * i + 2] = (px[i] >> 16) & 255;
            }
        }
    }
}

// k_warp_nearest -- the fused warp with cv::remap's INTER_NEAREST (FrameSourceWarp.hpp:90 admits the flag, :311 passes it
// on; the reference itself only ever passes INTER_LINEAR): createMap.cl's map, cvRound (half to even) + saturate_cast<short>
// per coordinate, ONE tap converted with the cvtColor arithmetic, 0 outside the source.  Direct gather: this mode is
// about completeness, the bilinear kernel carries the rate.
template <bool OCL>  // OCL: the map in the arithmetic of the reference's kernel as built for this GPU (VSTAB_MAP_CREATEMAP_CL_OPENCL)
__global__ void __launch_bounds__(256) k_warp_nearest(WarpArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.dw || y >= a.dh) return;
    float mx, my;
    if constexpr (OCL) {
        const float vx = ocl_div((float)x - a.p.ocx, a.p.ofx), vy = ocl_div((float)y - a.p.ocy, a.p.ofy);
        map_pixel_ocl_literal(a.p.icx, a.p.icy, a.p.ifx, a.p.ify, a.p.r, a.p.r[0] * vx, a.p.r[3] * vx, a.p.r[6] * vx, vy, mx, my);
    } else {
        map_pixel(a.p, col_term(a.p, x), row_term(a.p, y), mx, my);
    }
    // cvRound -> int (NaN / out of range: INT_MIN on x86), then saturate_cast<short>
    const bool far = !(fabsf(mx) < 2147483520.0f) || !(fabsf(my) < 2147483520.0f);
    const int sx = far ? -32768 : min(max((int)__builtin_rintf(mx), -32768), 32767), sy = far ? -32768 : min(max((int)__builtin_rintf(my), -32768), 32767);
    int b, g, r;
    fetch_tap(a, sx, sy, b, g, r);
    uint8_t *o = a.dst + (size_t)y * a.pitch_dst + (size_t)x * 3;
    o[0] = (uint8_t)b, o[1] = (uint8_t)g, o[2] = (uint8_t)r;
}

// k_quantised_map -- the map of every output pixel as cv::remap quantises it (32 * map rounded half to even; a NaN
// entry gets x = INT_MIN), written once for a run of frames that share their warp parameters (see CACHED above).
// The map arithmetic of k_warp_fused; its cvRound is the plain formulation (rint, NaN -> INT_MIN).
template <int MODE>
__global__ void __launch_bounds__(256) k_quantised_map(int2 *__restrict__ qmap, int qpitch, int dw, int dh, MapParams p, MapParams32 p32) {
    const int x0 = (blockIdx.x * 16 + (threadIdx.x & 15)) * 4, y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x0 >= qpitch || y >= dh) return;
    const float rfx = rcp_refined(p.ofx), rfy = rcp_refined(p.ofy);
    const float vy = norm_coord<MODE>((float)y - p.ocy, p.ofy, rfy);
    const RowTerm rt = {p.r[1] * vy, p.r[4] * vy, p.r[7] * vy};
    int q[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float vx = norm_coord<MODE>((float)(x0 + i) - p.ocx, p.ofx, rfx);
        const ColTerm ct = {p.r[0] * vx, p.r[3] * vx, p.r[6] * vx};
        float ax, ay;
        map_pixel_ex<MODE>(p32, p, ct, rt, vx, vy, ax, ay);
        const bool ok = !__builtin_isunordered(ax, ay);
        q[2 * i] = ok ? (int)__builtin_rintf(ax) : INT_MIN, q[2 * i + 1] = (int)__builtin_rintf(ay);
    }
    uint4 *o = reinterpret_cast<uint4 *>(qmap + (size_t)y * qpitch + x0);
    o[0] = make_uint4((uint32_t)q[0], (uint32_t)q[1], (uint32_t)q[2], (uint32_t)q[3]);
    o[1] = make_uint4((uint32_t)q[4], (uint32_t)q[5], (uint32_t)q[6], (uint32_t)q[7]);
}

// k_draw_markers -- the debug overlay of the lens surface (libdewobble's `debug` option, render.ts:678): a filled
// (2 * half + 1)^2 square at every tracked feature, clipped to the image.  One workgroup per feature.
__global__ void __launch_bounds__(64) k_draw_markers(uint8_t *__restrict__ dst, size_t pitch, int w, int h, int channels,
                                                     const int2 *__restrict__ centres, int half, uint32_t bgr) {
    const int2 c = centres[blockIdx.x];
    const int side = 2 * half + 1;
    for (int e = threadIdx.x; e < side * side; e += 64) {
        const int x = c.x - half + e % side, y = c.y - half + e / side;
        if (x < 0 || y < 0 || x >= w || y >= h) continue;
        uint8_t *o = dst + (size_t)y * pitch + (size_t)x * channels;
        o[0] = bgr & 255;
        if (channels == 3) o[1] = (bgr >> 8) & 255, o[2] = (bgr >> 16) & 255;
    }
}

static MapParams to_params(const float p[17]) {
    MapParams m;
    m.icx = p[0], m.icy = p[1], m.ifx = p[2], m.ify = p[3];
    m.ocx = p[4], m.ocy = p[5], m.ofx = p[6], m.ofy = p[7];
    for (int i = 0; i < 9; i++) m.r[i] = p[8 + i];
    return m;
}

static inline bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace vstab

using namespace vstab;

namespace vstab {
// luma_only: the 10-bit pipeline narrows the luma plane for the tracker and warps from the 16-bit planes
vstab_status pack_p010_planes(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int width, int height, void *dst, bool luma_only,
                                     void *stream) {
    if (!y || !uv || !dst) return fail(VSTAB_ERR_INVALID, "vstab_pack_p010: null pointer");
    if (width <= 0 || height <= 0 || (width & 1) || (height & 1)) return fail(VSTAB_ERR_INVALID, "Mismatched image dimensions");
    if (pitch_y < (size_t)width * 2 || pitch_uv < (size_t)width * 2) return fail(VSTAB_ERR_INVALID, "vstab_pack_p010: pitch smaller than row");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rows = luma_only ? height : height + height / 2;
    const bool vec = aligned(y, 16) && aligned(uv, 16) && aligned(dst, 8) && pitch_y % 16 == 0 && pitch_uv % 16 == 0 && width % 8 == 0;
    if (vec) {
        const int units = width / 8;
        dim3 grid(std::min<unsigned>(div_up((unsigned)((long)units * rows), 256), 2048));
        hipLaunchKernelGGL(k_pack_p010<true>, grid, dim3(256), 0, s, (const uint8_t *)y, pitch_y, (const uint8_t *)uv, pitch_uv, units, height, rows,
                           (uint8_t *)dst, (size_t)width);
    } else {
        dim3 grid(std::min<unsigned>(div_up((unsigned)((long)width * rows), 256), 2048));
        hipLaunchKernelGGL(k_pack_p010<false>, grid, dim3(256), 0, s, (const uint8_t *)y, pitch_y, (const uint8_t *)uv, pitch_uv, width, height, rows,
                           (uint8_t *)dst, (size_t)width);
    }
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}
}  // namespace vstab

// vstab_pack_nv12 with an optional event that completes with the copy kernel -- bound to the launch itself (hipExtLaunchKernelGGL's
// stop event), no marker packet on the stream: the pipeline waits for THAT before it lets upstream recycle a surface, not for the
// pyramid kernels enqueued behind the copy
namespace vstab {
vstab_status pack_nv12_planes(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int width, int height, void *dst, void *stream, hipEvent_t done) {
    if (!y || !uv || !dst) return fail(VSTAB_ERR_INVALID, "vstab_pack_nv12: null pointer");
    if (width <= 0 || height <= 0 || (width & 1) || (height & 1))
        return fail(VSTAB_ERR_INVALID, "Mismatched image dimensions");  // FrameSourceFfmpegOpenCl.cpp:53-56
    if (pitch_y < (size_t)width || pitch_uv < (size_t)width)
        return fail(VSTAB_ERR_INVALID, "vstab_pack_nv12: pitch smaller than row");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool v16 = aligned(y, 16) && aligned(uv, 16) && aligned(dst, 16) && pitch_y % 16 == 0 &&
                     pitch_uv % 16 == 0 && width % 16 == 0;
    const int rows = height + height / 2;
    if (v16) {
        const int vecs = width / 16;
        dim3 grid(std::min<unsigned>(div_up((unsigned)((long)vecs * rows), 256), 1024));
        hipExtLaunchKernelGGL(k_pack_nv12<uint4>, grid, dim3(256), 0, s, nullptr, done, 0, (const uint8_t *)y, pitch_y,
                           (const uint8_t *)uv, pitch_uv, vecs, height, (uint8_t *)dst, (size_t)width);
    } else if (aligned(y, 8) && aligned(uv, 8) && aligned(dst, 8) && pitch_y % 8 == 0 && pitch_uv % 8 == 0 && width % 8 == 0) {
        const int vecs = width / 8;
        dim3 grid(std::min<unsigned>(div_up((unsigned)((long)vecs * rows), 256), 1024));
        hipExtLaunchKernelGGL(k_pack_nv12<uint2>, grid, dim3(256), 0, s, nullptr, done, 0, (const uint8_t *)y, pitch_y,
                           (const uint8_t *)uv, pitch_uv, vecs, height, (uint8_t *)dst, (size_t)width);
    } else if (aligned(y, 4) && aligned(uv, 4) && aligned(dst, 4) && pitch_y % 4 == 0 && pitch_uv % 4 == 0 && width % 4 == 0) {
        const int vecs = width / 4;
        dim3 grid(std::min<unsigned>(div_up((unsigned)((long)vecs * rows), 256), 1024));
        hipExtLaunchKernelGGL(k_pack_nv12<uint32_t>, grid, dim3(256), 0, s, nullptr, done, 0, (const uint8_t *)y, pitch_y,
                           (const uint8_t *)uv, pitch_uv, vecs, height, (uint8_t *)dst, (size_t)width);
    } else {
        dim3 grid(std::min<unsigned>(div_up((unsigned)((long)width * rows), 256), 1024));
        hipExtLaunchKernelGGL(k_pack_nv12<uint8_t>, grid, dim3(256), 0, s, nullptr, done, 0, (const uint8_t *)y, pitch_y,
                           (const uint8_t *)uv, pitch_uv, width, height, (uint8_t *)dst, (size_t)width);
    }
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}

// Kernels of this translation unit are one code object, loaded by the runtime at the first launch of any of them.  Touching one of them
// here (vstab_preload_kernels) moves that load to a moment the caller chooses.
vstab_status preload_warp_kernels() {
    hipFuncAttributes at;
    VSTAB_HIP_TRY(hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&k_pack_nv12<uint32_t>)));
    return VSTAB_OK;
}

}  // namespace vstab

extern "C" {

vstab_status vstab_pack_nv12(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int width,
                             int height, void *dst, void *stream) {
    return pack_nv12_planes(y, pitch_y, uv, pitch_uv, width, height, dst, stream, nullptr);
}

vstab_status vstab_pack_p010(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int width, int height,
                             void *dst, void *stream) {
    return pack_p010_planes(y, pitch_y, uv, pitch_uv, width, height, dst, false, stream);
}

vstab_status vstab_cvt_nv12_bgr(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int width,
                                int height, void *dst, size_t pitch_dst, void *stream) {
    if (!y || !uv || !dst) return fail(VSTAB_ERR_INVALID, "vstab_cvt_nv12_bgr: null pointer");
    if (width <= 0 || height <= 0 || (width & 1) || (height & 1))
        return fail(VSTAB_ERR_INVALID, "vstab_cvt_nv12_bgr: width and height must be even");  // OpenCV asserts
    if (pitch_y < (size_t)width || pitch_uv < (size_t)width || pitch_dst < (size_t)width * 3)
        return fail(VSTAB_ERR_INVALID, "vstab_cvt_nv12_bgr: pitch smaller than row");
    const int vec_ok = aligned(y, 8) && aligned(uv, 8) && aligned(dst, 8) && pitch_y % 8 == 0 &&
                       pitch_uv % 8 == 0 && pitch_dst % 8 == 0;
    dim3 block(32, 8);
    dim3 grid(div_up(div_up(width, 8), 32), div_up(div_up(height, 2), 8));
    hipLaunchKernelGGL(k_cvt_nv12_bgr, grid, block, 0, static_cast<hipStream_t>(stream), (const uint8_t *)y,
                       pitch_y, (const uint8_t *)uv, pitch_uv, width, height, (uint8_t *)dst, pitch_dst, vec_ok);
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}

vstab_status vstab_create_map_ex(void *map_x, size_t pitch_x, void *map_y, size_t pitch_y, int cols, int rows,
                                 const float params[17], int map_mode, void *stream) {
    if (!map_x || !map_y || !params) return fail(VSTAB_ERR_INVALID, "vstab_create_map: null pointer");
    if (cols <= 0 || rows <= 0 || cols > 32767 || rows > 32767)
        return fail(VSTAB_ERR_INVALID, "vstab_create_map: size must be in [1, 32767] (createMap.cl:10-11)");
    if (pitch_x < (size_t)cols * 4 || pitch_y < (size_t)cols * 4 || pitch_x % 4 || pitch_y % 4)
        return fail(VSTAB_ERR_INVALID, "vstab_create_map: bad pitch");
    if (map_mode < VSTAB_MAP_CREATEMAP_CL || map_mode > VSTAB_MAP_CREATEMAP_CL_OPENCL) return fail(VSTAB_ERR_INVALID, "vstab_create_map: unknown map mode");
    const int vec_ok = aligned(map_x, 16) && aligned(map_y, 16) && pitch_x % 16 == 0 && pitch_y % 16 == 0;
    dim3 grid(div_up(div_up(cols, 4), 16), div_up(rows, 16));
    hipStream_t s = static_cast<hipStream_t>(stream);
#define VSTAB_LAUNCH(K) hipLaunchKernelGGL(K, grid, dim3(16, 16), 0, s, (float *)map_x, pitch_x, (float *)map_y, pitch_y, cols, rows, to_params(params), vec_ok)
    switch (map_mode) {
        case VSTAB_MAP_CREATEMAP_CL: VSTAB_LAUNCH(k_create_map); break;
        case VSTAB_MAP_FISH_TO_RECT: VSTAB_LAUNCH(k_create_map_ex<MAP_FISH_TO_RECT>); break;
        case VSTAB_MAP_FISH_TO_FISH: VSTAB_LAUNCH(k_create_map_ex<MAP_FISH_TO_FISH>); break;
        case VSTAB_MAP_RECT_TO_RECT: VSTAB_LAUNCH(k_create_map_ex<MAP_RECT_TO_RECT>); break;
        case VSTAB_MAP_RECT_TO_FISH: VSTAB_LAUNCH(k_create_map_ex<MAP_RECT_TO_FISH>); break;
        default: VSTAB_LAUNCH(k_create_map_ex<MAP_CREATEMAP_CL_OPENCL>); break;
    }
#undef VSTAB_LAUNCH
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}

vstab_status vstab_create_map(void *map_x, size_t pitch_x, void *map_y, size_t pitch_y, int cols, int rows,
                              const float params[17], void *stream) {
    return vstab_create_map_ex(map_x, pitch_x, map_y, pitch_y, cols, rows, params, VSTAB_MAP_CREATEMAP_CL, stream);
}

vstab_status vstab_remap_bilinear(const void *src, size_t pitch_src, int sw, int sh, int channels,
                                  const void *map_x, size_t pitch_x, const void *map_y, size_t pitch_y,
                                  void *dst, size_t pitch_dst, int dw, int dh, void *stream) {
    if (!src || !map_x || !map_y || !dst) return fail(VSTAB_ERR_INVALID, "vstab_remap_bilinear: null pointer");
    if (channels != 1 && channels != 3) return fail(VSTAB_ERR_INVALID, "vstab_remap_bilinear: channels must be 1 or 3");
    if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || sw > 32767 || sh > 32767)
        return fail(VSTAB_ERR_INVALID, "vstab_remap_bilinear: bad size");
    if (pitch_src < (size_t)sw * channels || pitch_dst < (size_t)dw * channels || pitch_x < (size_t)dw * 4 ||
        pitch_y < (size_t)dw * 4)
        return fail(VSTAB_ERR_INVALID, "vstab_remap_bilinear: pitch smaller than row");
    dim3 grid(div_up(dw, 64), div_up(dh, 4));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (channels == 3)
        hipLaunchKernelGGL(k_remap_bilinear<3>, grid, dim3(64, 4), 0, s, (const uint8_t *)src, pitch_src, sw, sh,
                           (const float *)map_x, pitch_x, (const float *)map_y, pitch_y, (uint8_t *)dst,
                           pitch_dst, dw, dh);
    else
        hipLaunchKernelGGL(k_remap_bilinear<1>, grid, dim3(64, 4), 0, s, (const uint8_t *)src, pitch_src, sw, sh,
                           (const float *)map_x, pitch_x, (const float *)map_y, pitch_y, (uint8_t *)dst,
                           pitch_dst, dw, dh);
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}

}  // extern "C"

static vstab_status warp_impl(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int sw, int sh,
                              const float params[17], int map_mode, int out_format, void *dst, size_t pitch_dst,
                              void *dst_uv, size_t pitch_dst_uv, int dw, int dh, void *stream, const void *qmap, int qpitch,
                              const float *rot_bottom = nullptr) {
    if (!y || !uv || !dst || !params) return fail(VSTAB_ERR_INVALID, "vstab_warp_nv12: null pointer");
    if (sw <= 0 || sh <= 0 || (sw & 1) || (sh & 1) || sw > 32767 || sh > 32767)
        return fail(VSTAB_ERR_INVALID, "vstab_warp_nv12: source must be even-sized and <= 32767");
    if (dw <= 0 || dh <= 0 || dw > 32767 || dh > 32767)
        return fail(VSTAB_ERR_INVALID, "vstab_warp_nv12: output size must be in [1, 32767]");
    if (map_mode < VSTAB_MAP_CREATEMAP_CL || map_mode > VSTAB_MAP_CREATEMAP_CL_OPENCL) return fail(VSTAB_ERR_INVALID, "vstab_warp_nv12: unknown map mode");
    if (out_format != VSTAB_OUT_BGR8 && out_format != VSTAB_OUT_NV12 && out_format != VSTAB_OUT_NV12_PLANAR)
        return fail(VSTAB_ERR_INVALID, "vstab_warp_nv12: unknown output format");
    const bool planar = out_format == VSTAB_OUT_NV12_PLANAR;
    const bool nv12_out = out_format == VSTAB_OUT_NV12 || planar;
    if (pitch_y < (size_t)sw || pitch_uv < (size_t)sw || pitch_dst < (size_t)dw * (nv12_out ? 1 : 3))
        return fail(VSTAB_ERR_INVALID, "vstab_warp_nv12: pitch smaller than row");
    if (nv12_out && (!dst_uv || pitch_dst_uv < (size_t)((dw + 1) / 2) * 2))
        return fail(VSTAB_ERR_INVALID, "vstab_warp_nv12: NV12 output needs a chroma plane of 2*ceil(width/2) bytes per row");
    if (!aligned(uv, 2) || pitch_uv % 2) return fail(VSTAB_ERR_INVALID, "vstab_warp_nv12: chroma plane must be 2-B aligned");
    WarpArgs a;
    a.y = (const uint8_t *)y, a.uv = (const uint8_t *)uv, a.dst = (uint8_t *)dst, a.dst_uv = (uint8_t *)dst_uv;
    a.pitch_y = pitch_y, a.pitch_uv = pitch_uv, a.pitch_dst = pitch_dst, a.pitch_dst_uv = pitch_dst_uv;
    a.sw = sw, a.sh = sh, a.dw = dw, a.dh = dh;
    a.p = to_params(params);
    const int vec_ok = aligned(dst, 4) && pitch_dst % 4 == 0 && (!nv12_out || (aligned(dst_uv, 4) && pitch_dst_uv % 4 == 0));
    const bool small_pitch = pitch_y < (1u << 24) && pitch_uv < (1u << 24) && (uint64_t)pitch_y * sh < (1ull << 32);
    if (rot_bottom && map_mode != VSTAB_MAP_CREATEMAP_CL && map_mode != VSTAB_MAP_FISH_TO_RECT && map_mode != VSTAB_MAP_CREATEMAP_CL_OPENCL)
        return fail(VSTAB_ERR_INVALID, "vstab_warp_nv12_rs: the per-row warp exists for the fisheye -> pinhole modes (0, 1, 5) only");
    const bool plain = map_mode == VSTAB_MAP_CREATEMAP_CL && !nv12_out && !qmap && !rot_bottom;
    if (!plain && !small_pitch) return fail(VSTAB_ERR_INVALID, "vstab_warp_nv12: source pitch too large for this mode");
    if (planar) {  // the plane-wise warp: its own kernel (vstab_warp_planar.hip); the map is always evaluated
        if (qmap) return fail(VSTAB_ERR_UNSUPPORTED, "vstab_warp_nv12_mapped: the quantised map holds no chroma positions -- VSTAB_OUT_NV12_PLANAR goes through vstab_warp_nv12_ex");
        if (sw < 16 || sh < 2) return fail(VSTAB_ERR_INVALID, "vstab_warp_nv12: the plane-wise warp needs a source of at least 16 x 2");
        const bool src16 = aligned(y, 16) && aligned(uv, 16) && pitch_y % 16 == 0 && pitch_uv % 16 == 0;  // 16-byte staging loads
        const bool dst16 = aligned(dst, 16) && aligned(dst_uv, 16) && pitch_dst % 16 == 0 && pitch_dst_uv % 16 == 0;
        return launch_warp_planar(a, params, map_mode, 8, 0, src16, dst16, rot_bottom, static_cast<hipStream_t>(stream));
    }
    bool direct = !small_pitch;
#ifdef VSTAB_DEV
    static const int variant = getenv("VSTAB_WARP_VARIANT") ? atoi(getenv("VSTAB_WARP_VARIANT")) : 2;
    direct = direct || (plain && variant == 1);
#endif
    if (direct) {  // frames of 4 GiB and more: the direct-gather kernel with 64-bit addressing
        dim3 grid(div_up(dw, WARP_TILE_W), div_up(dh, WARP_TILE_H));
        hipLaunchKernelGGL(k_warp_nv12_bgr, grid, dim3(16, 16), 0, static_cast<hipStream_t>(stream), a, vec_ok);
    } else {
        const bool src_vec_ok = aligned(y, 8) && aligned(uv, 8) && pitch_y % 8 == 0 && pitch_uv % 8 == 0;  // 8-byte staging loads
        return launch_warp_fused(a, params, map_mode, nv12_out, src_vec_ok, vec_ok, qmap, qpitch, rot_bottom, static_cast<hipStream_t>(stream));
    }
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}

vstab_status vstab_draw_markers(void *dst, size_t pitch, int width, int height, int channels, const int *centres_xy, int n, int half,
                                unsigned int bgr, void *stream) {
    if (!dst || width <= 0 || height <= 0 || (channels != 1 && channels != 3) || pitch < (size_t)width * channels || n < 0 || half < 0 ||
        half > 16 || (n > 0 && !centres_xy))
        return fail(VSTAB_ERR_INVALID, "vstab_draw_markers: bad argument");
    if (n == 0) return VSTAB_OK;
    hipLaunchKernelGGL(k_draw_markers, dim3(n), dim3(64), 0, static_cast<hipStream_t>(stream), (uint8_t *)dst, pitch, width, height, channels,
                       reinterpret_cast<const int2 *>(centres_xy), half, bgr);
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}

extern "C" {

vstab_status vstab_warp_nv12_ex(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int sw, int sh,
                                const float params[17], int map_mode, int out_format, void *dst, size_t pitch_dst,
                                void *dst_uv, size_t pitch_dst_uv, int dw, int dh, void *stream) {
    return warp_impl(y, pitch_y, uv, pitch_uv, sw, sh, params, map_mode, out_format, dst, pitch_dst, dst_uv, pitch_dst_uv, dw, dh, stream, nullptr, 0);
}

vstab_status vstab_warp_nv12_rs(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int sw, int sh, const float params[17],
                                const float rot_bottom[9], int map_mode, int out_format, void *dst, size_t pitch_dst, void *dst_uv,
                                size_t pitch_dst_uv, int dw, int dh, void *stream) {
    if (!rot_bottom) return fail(VSTAB_ERR_INVALID, "vstab_warp_nv12_rs: null pointer");
    return warp_impl(y, pitch_y, uv, pitch_uv, sw, sh, params, map_mode, out_format, dst, pitch_dst, dst_uv, pitch_dst_uv, dw, dh, stream, nullptr, 0,
                     rot_bottom);
}

size_t vstab_quantised_map_bytes(int dst_width, int dst_height) {
    return dst_width > 0 && dst_height > 0 ? (size_t)((dst_width + 3) & ~3) * dst_height * 8 : 0;
}

vstab_status vstab_quantised_map(void *qmap, int dw, int dh, const float params[17], int map_mode, void *stream) {
    if (!qmap || !params || dw <= 0 || dh <= 0 || dw > 32767 || dh > 32767) return fail(VSTAB_ERR_INVALID, "vstab_quantised_map: bad argument");
    if (map_mode < VSTAB_MAP_CREATEMAP_CL || map_mode > VSTAB_MAP_CREATEMAP_CL_OPENCL) return fail(VSTAB_ERR_INVALID, "vstab_quantised_map: unknown map mode");
    if (!aligned(qmap, 16)) return fail(VSTAB_ERR_INVALID, "vstab_quantised_map: the buffer must be 16-byte aligned");
    const int qpitch = (dw + 3) & ~3;
    const MapParams p = to_params(params);
    const MapParams32 p32 = {params[0] * 32.0f, params[1] * 32.0f, params[2] * 32.0f, params[3] * 32.0f, params[10], params[13], params[16]};
    dim3 grid(div_up(qpitch / 4, 16), div_up(dh, 16));
    hipStream_t st = static_cast<hipStream_t>(stream);
#define VSTAB_LAUNCH(M) hipLaunchKernelGGL(k_quantised_map<M>, grid, dim3(256), 0, st, static_cast<int2 *>(qmap), qpitch, dw, dh, p, p32)
    switch (map_mode) {
        case VSTAB_MAP_CREATEMAP_CL: VSTAB_LAUNCH(MAP_CREATEMAP_CL); break;
        case VSTAB_MAP_FISH_TO_RECT: VSTAB_LAUNCH(MAP_FISH_TO_RECT); break;
        case VSTAB_MAP_FISH_TO_FISH: VSTAB_LAUNCH(MAP_FISH_TO_FISH); break;
        case VSTAB_MAP_RECT_TO_RECT: VSTAB_LAUNCH(MAP_RECT_TO_RECT); break;
        case VSTAB_MAP_RECT_TO_FISH: VSTAB_LAUNCH(MAP_RECT_TO_FISH); break;
        default: VSTAB_LAUNCH(MAP_CREATEMAP_CL_OPENCL); break;
    }
#undef VSTAB_LAUNCH
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}

vstab_status vstab_warp_nv12_mapped(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int sw, int sh, const void *qmap,
                                    int out_format, void *dst, size_t pitch_dst, void *dst_uv, size_t pitch_dst_uv, int dw, int dh,
                                    void *stream) {
    if (!qmap || !aligned(qmap, 16)) return fail(VSTAB_ERR_INVALID, "vstab_warp_nv12_mapped: the quantised map must be a 16-byte aligned device buffer");
    static const float unused[17] = {0};
    return warp_impl(y, pitch_y, uv, pitch_uv, sw, sh, unused, VSTAB_MAP_CREATEMAP_CL, out_format, dst, pitch_dst, dst_uv, pitch_dst_uv, dw, dh, stream,
                     qmap, (dw + 3) & ~3);
}

vstab_status vstab_warp_nv12_nearest_ex(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int sw, int sh, const float params[17],
                                        int map_mode, void *dst, size_t pitch_dst, int dw, int dh, void *stream) {
    if (!y || !uv || !dst || !params) return fail(VSTAB_ERR_INVALID, "vstab_warp_nv12_nearest: null pointer");
    if (sw <= 0 || sh <= 0 || (sw & 1) || (sh & 1) || sw > 32767 || sh > 32767 || dw <= 0 || dh <= 0 || dw > 32767 || dh > 32767)
        return fail(VSTAB_ERR_INVALID, "vstab_warp_nv12_nearest: sizes must be in [1, 32767], source even");
    if (pitch_y < (size_t)sw || pitch_uv < (size_t)sw || pitch_dst < (size_t)dw * 3 || !aligned(uv, 2) || pitch_uv % 2)
        return fail(VSTAB_ERR_INVALID, "vstab_warp_nv12_nearest: bad pitch or chroma alignment");
    if (map_mode != VSTAB_MAP_CREATEMAP_CL && map_mode != VSTAB_MAP_CREATEMAP_CL_OPENCL)
        return fail(VSTAB_ERR_INVALID, "vstab_warp_nv12_nearest: the nearest-neighbour warp exists for the reference's own map (modes 0 and 5)");
    WarpArgs a;
    a.y = (const uint8_t *)y, a.uv = (const uint8_t *)uv, a.dst = (uint8_t *)dst, a.dst_uv = nullptr;
    a.pitch_y = pitch_y, a.pitch_uv = pitch_uv, a.pitch_dst = pitch_dst, a.pitch_dst_uv = 0;
    a.sw = sw, a.sh = sh, a.dw = dw, a.dh = dh;
    a.p = to_params(params);
    const dim3 grid(div_up(dw, 64), div_up(dh, 4));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const LaunchEvents ev = take_launch_events();  // a profiling caller's pair: the kernel's own start / end stamps
    if (map_mode == VSTAB_MAP_CREATEMAP_CL_OPENCL) {
        if (ev.start) hipExtLaunchKernelGGL(k_warp_nearest<true>, grid, dim3(256), 0, st, ev.start, ev.stop, 0, a);
        else hipLaunchKernelGGL(k_warp_nearest<true>, grid, dim3(256), 0, st, a);
    } else {
        if (ev.start) hipExtLaunchKernelGGL(k_warp_nearest<false>, grid, dim3(256), 0, st, ev.start, ev.stop, 0, a);
        else hipLaunchKernelGGL(k_warp_nearest<false>, grid, dim3(256), 0, st, a);
    }
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}

vstab_status vstab_warp_nv12_nearest(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int sw, int sh, const float params[17],
                                     void *dst, size_t pitch_dst, int dw, int dh, void *stream) {
    return vstab_warp_nv12_nearest_ex(y, pitch_y, uv, pitch_uv, sw, sh, params, VSTAB_MAP_CREATEMAP_CL, dst, pitch_dst, dw, dh, stream);
}

vstab_status vstab_warp_nv12_bgr(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int sw, int sh,
                                 const float params[17], void *dst, size_t pitch_dst, int dw, int dh,
                                 void *stream) {
    return vstab_warp_nv12_ex(y, pitch_y, uv, pitch_uv, sw, sh, params, VSTAB_MAP_CREATEMAP_CL, VSTAB_OUT_BGR8, dst, pitch_dst,
                              nullptr, 0, dw, dh, stream);
}

}  // extern "C"
