// vstab_warp_args.hpp -- kernel argument blocks and the direct-gather sampling shared by the warp kernels.
#pragma once
#include "vstab_device.hpp"
#include "vstab_internal.hpp"

namespace vstab {

struct WarpArgs {
    const uint8_t *y;
    const uint8_t *uv;
    uint8_t *dst;     // BGR8, or the luma plane in NV12 output mode
    uint8_t *dst_uv;  // NV12 output mode: interleaved chroma plane, ceil(dh/2) rows of 2*ceil(dw/2) bytes
    size_t pitch_y, pitch_uv, pitch_dst, pitch_dst_uv;
    int sw, sh, dw, dh;
    MapParams p;
};

struct FusedArgs {
    WarpArgs w;
    MapParams32 p32;
    int lds_capacity_px;  // dwords available for the staged tile
    int src_vec_ok;       // planes and pitches 8-B aligned -> 8-byte staging loads
    int dst_vec_ok;
    int tiles_x, tiles_y;
    const int2 *qmap;  // CACHED kernels: the quantised map (32 * map rounded to int, NaN -> INT_MIN in x) of every output pixel
    int qpitch;        // its row pitch in pixels (a multiple of 4)
    float rs_d[9];     // rolling-shutter modes: rotation of the last output row minus rotation of the first (w.p.r), fp32
    float rs_den;      // and (float)max(dh - 1, 1)
    int band_y[9];     // XCD k (= blockIdx % 8) owns output rows [band_y[k], band_y[k + 1])
    int split_y[8];    // in its band: rows below split_y[k] in tall tiles (4 RWB rows), from it on in half-height tiles
#ifdef VSTAB_DEV
    unsigned long long *timing;  // development builds: 4 qwords per workgroup {memrealtime, memtime at entry and exit}
    int ablate;                  // timing-only ablations (wrong pixels): 1 linear map, 2 xor blend, 4 raw conversion, 8 no stores
    int lds_pad;                 // experiment: dwords added to the LDS row pitch of the staged box (multiple of 4)
#endif
};

// One NV12 tap converted with the cvtColor arithmetic; outside the source -> 0 (cv::remap BORDER_CONSTANT).
__device__ __forceinline__ void fetch_tap(const WarpArgs &a, int X, int Y, int &b, int &g, int &r) {
    if ((unsigned)X < (unsigned)a.sw && (unsigned)Y < (unsigned)a.sh) {
        const int yv = a.y[(size_t)Y * a.pitch_y + X];
        const uint16_t c = *reinterpret_cast<const uint16_t *>(a.uv + (size_t)(Y >> 1) * a.pitch_uv + (X & ~1));
        yuv_to_bgr(yv, chroma_term(c & 255, c >> 8), b, g, r);
    } else {
        b = g = r = 0;
    }
}

// cv::remap's four-product fixed-point blend of taps fetched straight from global memory.  sx, sy = rint(32 * map).
__device__ __forceinline__ uint32_t gather_pixel(const WarpArgs &a, int sx, int sy) {
    const int X = sx >> 5, Y = sy >> 5;
    const uint32_t fx = sx & 31, fy = sy & 31;
    const int w00 = (32 - fx) * (32 - fy), w01 = fx * (32 - fy), w10 = (32 - fx) * fy, w11 = fx * fy;
    int b0, g0, r0, b1, g1, r1, b2, g2, r2, b3, g3, r3;
    fetch_tap(a, X, Y, b0, g0, r0);
    fetch_tap(a, X + 1, Y, b1, g1, r1);
    fetch_tap(a, X, Y + 1, b2, g2, r2);
    fetch_tap(a, X + 1, Y + 1, b3, g3, r3);
    const uint32_t B = (uint32_t)(b0 * w00 + b1 * w01 + b2 * w10 + b3 * w11 + 512) >> 10;
    const uint32_t G = (uint32_t)(g0 * w00 + g1 * w01 + g2 * w10 + g3 * w11 + 512) >> 10;
    const uint32_t R = (uint32_t)(r0 * w00 + r1 * w01 + r2 * w10 + r3 * w11 + 512) >> 10;
    return B | (G << 8) | (R << 16);
}

__device__ __forceinline__ uint32_t load_u32_bytes(const uint8_t *p, int valid) {
    uint32_t v = 0;
    for (int i = 0; i < 4; i++)
        if (i < valid) v |= (uint32_t)p[i] << (8 * i);
    return v;
}

vstab_status launch_warp_fused(const WarpArgs &a, const float params[17], int map_mode, bool nv12_out, bool src_vec_ok, bool dst_vec_ok,
                               const void *qmap, int qpitch, const float *rot_bottom, hipStream_t st);
// the 10-bit pixel path on the LDS-tiled kernel: a.y / a.uv = P010 planes (16-byte aligned, pitches multiples of 16), a.dst = 16-bit
// BGR with a.pitch_dst bytes per row; map modes 0 / 1 only
vstab_status launch_warp_fused10(const WarpArgs &a, const float params[17], int map_mode, int blend, const float *rot_bottom, bool p010_out, bool dst_vec_ok,
                                 hipStream_t st);

// the plane-wise warp (vstab_warp_planar.hip): a.dst / a.dst_uv = the output planes; depth 8 (NV12 bytes) or 10 (P010 words)
vstab_status launch_warp_planar(const WarpArgs &a, const float params[17], int map_mode, int depth, int blend, bool src_vec_ok, bool dst_vec_ok,
                                const float *rot_bottom, hipStream_t st);

}  // namespace vstab
