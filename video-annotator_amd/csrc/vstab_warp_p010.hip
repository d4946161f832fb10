// vstab_warp_p010.hip -- BASELINE.json config 5: the undistort-remap with a 10-bit pixel path.
// No reference counterpart (the reference is 8-bit throughout): the arithmetic is DEFINED in the oracle
// (its warp_p010 chain) and reproduced here bit for bit.
//   P010 planes (16-bit samples, 10 significant bits at the top)  ->  sample >> 6
//   BT.601 limited range at 10 bits: the 8-bit cvtColor constants and shift with offsets 64 / 512 (the definition sums in
//     64 bits: 959 * CY + 511 * CUB exceeds int32)  ->  B, G, R in [0, 1023]
//   the map of vstab_create_map_ex (all five projection pairs), optionally with a rotation per output row
//     (vstab_warp_nv12_rs's interpolation), quantised like cv::remap
//   blend of the four converted taps: VSTAB_BLEND_EXACT  (sum p*w + 512) >> 10, integers, as the 8-bit path;
//     VSTAB_BLEND_FP16  four fused multiply-adds in binary16 with the weights as w / 1024 (exact in binary16),
//     taps in the order 00, 01, 10, 11, then round-to-nearest-even and clamp -- the "fp16 blend" of config 5,
//     within 2 levels of the exact one
//   BGR, three 16-bit samples per pixel, values 0..1023.
// Direct gather, two output pixels per thread: this path is about the format, the 8-bit kernel carries the rate.
#include <cstdlib>

#include <hip/hip_ext.h>

#include "vstab_device10.hpp"
#include "vstab_internal.hpp"
#include "vstab_warp_args.hpp"

namespace vstab {

struct P010Args {
    const uint8_t *y, *uv;
    uint16_t *dst;
    size_t pitch_y, pitch_uv, pitch_dst;  // bytes
    int sw, sh, dw, dh;
    MapParams p;
    float rs_d[9];  // rotation of the last output row minus rotation of the first (all zero: one rotation)
    float rs_den;   // (float)max(dh - 1, 1)
    int rs;
};

// the four taps of one output pixel, converted and blended; sx, sy = cvRound(32 * map)
template <int BLEND>
__device__ __forceinline__ void sample10(const P010Args &a, float ax, float ay, uint16_t *o) {
    const bool far = !(fabsf(ax) < 1073741824.0f) || !(fabsf(ay) < 1073741824.0f);
    const int sx = (int)__builtin_rintf(ax), sy = (int)__builtin_rintf(ay);
    const int X = sx >> 5, Y = sy >> 5, fx = sx & 31, fy = sy & 31;
    int B = 0, G = 0, R = 0;
    if (!(far || X >= a.sw || X + 1 < 0 || Y >= a.sh || Y + 1 < 0)) {
        const int w00 = (32 - fx) * (32 - fy), w01 = fx * (32 - fy), w10 = (32 - fx) * fy, w11 = fx * fy;
        int b0, g0, r0, b1, g1, r1, b2, g2, r2, b3, g3, r3;
        fetch_row10(a, X, Y, b0, g0, r0, b1, g1, r1);
        fetch_row10(a, X, Y + 1, b2, g2, r2, b3, g3, r3);
        if constexpr (BLEND == VSTAB_BLEND_FP16) {
            B = blend_fp16(b0, b1, b2, b3, w00, w01, w10, w11);
            G = blend_fp16(g0, g1, g2, g3, w00, w01, w10, w11);
            R = blend_fp16(r0, r1, r2, r3, w00, w01, w10, w11);
        } else {
            B = (b0 * w00 + b1 * w01 + b2 * w10 + b3 * w11 + 512) >> 10;
            G = (g0 * w00 + g1 * w01 + g2 * w10 + g3 * w11 + 512) >> 10;
            R = (r0 * w00 + r1 * w01 + r2 * w10 + r3 * w11 + 512) >> 10;
        }
    }
    o[0] = (uint16_t)B, o[1] = (uint16_t)G, o[2] = (uint16_t)R;
}

// A thread owns two vertically adjacent output pixels, so that the map of the fisheye-input modes runs as packed fp32
// (map_pixel32_x2: the hand-scheduled IEEE sequences of the 8-bit kernel, two pixels per instruction; 32 * map, an
// exact power-of-two scaling of the map the definition quantises).
template <int MODE, int BLEND>
__global__ void __launch_bounds__(256) k_warp_p010(P010Args a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 2;
    if (x >= a.dw || y0 >= a.dh) return;
    float m[2][9];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const float t = (float)(y0 + r) / a.rs_den;
#pragma unroll
        for (int k = 0; k < 9; k++) m[r][k] = a.rs ? __builtin_fmaf(t, a.rs_d[k], a.p.r[k]) : a.p.r[k];
    }
    constexpr bool OCL = MODE == MAP_CREATEMAP_CL_OPENCL;  // createMap.cl as ROCm's OpenCL compiler builds it: reciprocal-based divisions
    const float vx = OCL ? ocl_div((float)x - a.p.ocx, a.p.ofx) : ((float)x - a.p.ocx) / a.p.ofx;
    const float vy[2] = {OCL ? ocl_div((float)y0 - a.p.ocy, a.p.ofy) : ((float)y0 - a.p.ocy) / a.p.ofy,
                         OCL ? ocl_div((float)(y0 + 1) - a.p.ocy, a.p.ofy) : ((float)(y0 + 1) - a.p.ocy) / a.p.ofy};
    float ax[2], ay[2];
    if constexpr (MODE == MAP_CREATEMAP_CL || MODE == MAP_FISH_TO_RECT) {
        f32x2 px, py;
        map_pixel32_x2<MODE == MAP_FISH_TO_RECT>(a.p.icx * 32.0f, a.p.icy * 32.0f, a.p.ifx * 32.0f, a.p.ify * 32.0f, (f32x2){m[0][2], m[1][2]},
                                                 (f32x2){m[0][5], m[1][5]}, (f32x2){m[0][8], m[1][8]}, (f32x2){m[0][0] * vx, m[1][0] * vx},
                                                 (f32x2){m[0][3] * vx, m[1][3] * vx}, (f32x2){m[0][6] * vx, m[1][6] * vx},
                                                 (f32x2){m[0][1] * vy[0], m[1][1] * vy[1]}, (f32x2){m[0][4] * vy[0], m[1][4] * vy[1]},
                                                 (f32x2){m[0][7] * vy[0], m[1][7] * vy[1]}, px, py);
        ax[0] = px.x, ax[1] = px.y, ay[0] = py.x, ay[1] = py.y;
    } else {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            MapParams q = a.p;
#pragma unroll
            for (int k = 0; k < 9; k++) q.r[k] = m[r][k];
            const ColTerm ct = {q.r[0] * vx, q.r[3] * vx, q.r[6] * vx};
            const RowTerm rt = {q.r[1] * vy[r], q.r[4] * vy[r], q.r[7] * vy[r]};
            const MapParams32 in = {q.icx, q.icy, q.ifx, q.ify, q.r[2], q.r[5], q.r[8]};  // unscaled
            float mx, my;
            map_pixel_ex<MODE>(in, q, ct, rt, vx, vy[r], mx, my);
            ax[r] = mx * 32.0f, ay[r] = my * 32.0f;
        }
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (y0 + r >= a.dh) break;
        sample10<BLEND>(a, ax[r], ay[r], reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(a.dst) + (size_t)(y0 + r) * a.pitch_dst) + 3 * (size_t)x);
    }
}

// k_cvt_bgr10_p010 -- the encoder hand-off of the 10-bit path: BGR (0..1023 in 16-bit containers) -> P010 planes, the BGR -> YUV
// arithmetic of the NV12 output (cvtColor's BT.601 constants, 20-bit shift) at 10 bits: offsets 64 / 512, saturation to
// [0, 1023], sample << 6; chroma from the top-left pixel of each 2 x 2 block (definition: include/vstab.h).  One
// thread converts two adjacent pixels of a row: 12 bytes in, one luma dword out, on even rows one (U, V) dword.
__global__ void __launch_bounds__(256) k_cvt_bgr10_p010(const uint8_t *__restrict__ src, size_t pitch_src, int w, int h, uint8_t *__restrict__ dy,
                                                        size_t pitch_y, uint8_t *__restrict__ duv, size_t pitch_uv) {
    const int px = (blockIdx.x * 64 + (threadIdx.x & 63)) * 2, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= w || y >= h) return;
    const uint16_t *s = reinterpret_cast<const uint16_t *>(src + (size_t)y * pitch_src) + 3 * (size_t)px;
    const bool two = px + 1 < w;
    const int B0 = s[0], G0 = s[1], R0 = s[2], B1 = two ? s[3] : 0, G1 = two ? s[4] : 0, R1 = two ? s[5] : 0;
    constexpr int half = 1 << 19;
    const int y0 = sat10((CRY * R0 + CGY * G0 + CBY * B0 + half + (64 << 20)) >> 20), y1 = sat10((CRY * R1 + CGY * G1 + CBY * B1 + half + (64 << 20)) >> 20);
    uint16_t *oy = reinterpret_cast<uint16_t *>(dy + (size_t)y * pitch_y) + px;
    if (two && (reinterpret_cast<uintptr_t>(oy) & 3) == 0) *reinterpret_cast<uint32_t *>(oy) = ((uint32_t)y0 << 6) | ((uint32_t)y1 << 22);
    else {
        oy[0] = (uint16_t)(y0 << 6);
        if (two) oy[1] = (uint16_t)(y1 << 6);
    }
    if (!(y & 1)) {
        const int U = sat10((CRU * R0 + CGU * G0 + CBU * B0 + half + (512 << 20)) >> 20), V = sat10((CBU * R0 + CGV * G0 + CBV * B0 + half + (512 << 20)) >> 20);
        *reinterpret_cast<uint32_t *>(duv + (size_t)(y >> 1) * pitch_uv + (size_t)px * 2) = ((uint32_t)U << 6) | ((uint32_t)V << 22);
    }
}


// Kernels of this translation unit are one code object, loaded by the runtime at the first launch of any of them.  Touching one of them
// here (vstab_preload_kernels) moves that load to a moment the caller chooses.
vstab_status preload_p010_kernels() {
    hipFuncAttributes at;
    VSTAB_HIP_TRY(hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&k_warp_p010<MAP_CREATEMAP_CL, VSTAB_BLEND_EXACT>)));
    return VSTAB_OK;
}

}  // namespace vstab

using namespace vstab;

// The same warp with P010 planes out, fused (FMT 3 of the tiled kernel): VSTAB_ERR_UNSUPPORTED when the planes do not allow the tiled
// kernel (the caller then warps to 16-bit BGR and converts: vstab_pull_frame_p010 does).
extern "C" vstab_status vstab_warp_p010_planes(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int sw, int sh, const float params[17],
                                               const float *rot_bottom, int map_mode, int blend, void *dst_y, size_t pitch_dst_y, void *dst_uv,
                                               size_t pitch_dst_uv, int dw, int dh, void *stream) {
    if (!y || !uv || !dst_y || !dst_uv || !params) return fail(VSTAB_ERR_INVALID, "vstab_warp_p010_planes: null pointer");
    if (sw < 8 || sh < 2 || (sw & 1) || (sh & 1) || dw <= 0 || dh <= 0 || sw > 32767 || sh > 32767 || dw > 32767 || dh > 32767)
        return fail(VSTAB_ERR_INVALID, "vstab_warp_p010_planes: sizes must be in [1, 32767], source even and at least 8 x 2");
    if (pitch_dst_y < (size_t)dw * 2 || pitch_dst_y % 2 || pitch_dst_uv < (size_t)((dw + 1) / 2) * 4 || pitch_dst_uv % 4 ||
        reinterpret_cast<uintptr_t>(dst_y) % 2 || reinterpret_cast<uintptr_t>(dst_uv) % 4)
        return fail(VSTAB_ERR_INVALID, "vstab_warp_p010_planes: bad output pitch or alignment (16-bit samples; chroma pairs 4-byte aligned)");
    if (blend != VSTAB_BLEND_EXACT && blend != VSTAB_BLEND_FP16) return fail(VSTAB_ERR_INVALID, "vstab_warp_p010_planes: unknown blend");
    const bool tiled_ok = (map_mode == VSTAB_MAP_CREATEMAP_CL || map_mode == VSTAB_MAP_FISH_TO_RECT || map_mode == VSTAB_MAP_CREATEMAP_CL_OPENCL) &&
                          reinterpret_cast<uintptr_t>(y) % 16 == 0 &&
                          reinterpret_cast<uintptr_t>(uv) % 16 == 0 && pitch_y % 16 == 0 && pitch_uv % 16 == 0 && pitch_y >= (size_t)sw * 2 &&
                          pitch_uv >= (size_t)sw * 2 && pitch_y < (1u << 24) && pitch_uv < (1u << 24) && (uint64_t)pitch_y * sh < (1ull << 32) &&
                          (uint64_t)pitch_dst_y * dh < (1ull << 32);
    if (!tiled_ok) return fail(VSTAB_ERR_UNSUPPORTED, "vstab_warp_p010_planes: needs 16-byte aligned source planes and a fisheye -> pinhole map");
    WarpArgs wa;
    wa.y = static_cast<const uint8_t *>(y), wa.uv = static_cast<const uint8_t *>(uv), wa.dst = static_cast<uint8_t *>(dst_y), wa.dst_uv = static_cast<uint8_t *>(dst_uv);
    wa.pitch_y = pitch_y, wa.pitch_uv = pitch_uv, wa.pitch_dst = pitch_dst_y, wa.pitch_dst_uv = pitch_dst_uv;
    wa.sw = sw, wa.sh = sh, wa.dw = dw, wa.dh = dh;
    wa.p = {params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7],
            {params[8], params[9], params[10], params[11], params[12], params[13], params[14], params[15], params[16]}};
    const bool vec = reinterpret_cast<uintptr_t>(dst_y) % 8 == 0 && reinterpret_cast<uintptr_t>(dst_uv) % 8 == 0 && pitch_dst_y % 8 == 0 && pitch_dst_uv % 8 == 0;
    return launch_warp_fused10(wa, params, map_mode, blend, rot_bottom, true, vec, static_cast<hipStream_t>(stream));
}

// The plane-wise 10-bit warp (vstab_warp_planar.hip, DEPTH 10): P010 planes in and out, no colour conversion at all.
extern "C" vstab_status vstab_warp_p010_planar(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int sw, int sh, const float params[17],
                                               const float *rot_bottom, int map_mode, int blend, void *dst_y, size_t pitch_dst_y, void *dst_uv,
                                               size_t pitch_dst_uv, int dw, int dh, void *stream) {
    if (!y || !uv || !dst_y || !dst_uv || !params) return fail(VSTAB_ERR_INVALID, "vstab_warp_p010_planar: null pointer");
    if (sw < 8 || sh < 2 || (sw & 1) || (sh & 1) || dw <= 0 || dh <= 0 || sw > 32767 || sh > 32767 || dw > 32767 || dh > 32767)
        return fail(VSTAB_ERR_INVALID, "vstab_warp_p010_planar: sizes must be in [1, 32767], source even and at least 8 x 2");
    if (pitch_y < (size_t)sw * 2 || pitch_uv < (size_t)sw * 2 || pitch_y % 2 || pitch_uv % 4 || reinterpret_cast<uintptr_t>(y) % 2 ||
        reinterpret_cast<uintptr_t>(uv) % 4)
        return fail(VSTAB_ERR_INVALID, "vstab_warp_p010_planar: bad source pitch or alignment (16-bit samples; chroma pairs 4-byte aligned)");
    if (pitch_dst_y < (size_t)dw * 2 || pitch_dst_y % 2 || pitch_dst_uv < (size_t)((dw + 1) / 2) * 4 || pitch_dst_uv % 4 ||
        reinterpret_cast<uintptr_t>(dst_y) % 2 || reinterpret_cast<uintptr_t>(dst_uv) % 4)
        return fail(VSTAB_ERR_INVALID, "vstab_warp_p010_planar: bad output pitch or alignment (16-bit samples; chroma pairs 4-byte aligned)");
    if (map_mode < VSTAB_MAP_CREATEMAP_CL || map_mode > VSTAB_MAP_CREATEMAP_CL_OPENCL) return fail(VSTAB_ERR_INVALID, "vstab_warp_p010_planar: unknown map mode");
    if (blend != VSTAB_BLEND_EXACT && blend != VSTAB_BLEND_FP16) return fail(VSTAB_ERR_INVALID, "vstab_warp_p010_planar: unknown blend");
    if (rot_bottom && map_mode != VSTAB_MAP_CREATEMAP_CL && map_mode != VSTAB_MAP_FISH_TO_RECT && map_mode != VSTAB_MAP_CREATEMAP_CL_OPENCL)
        return fail(VSTAB_ERR_INVALID, "vstab_warp_p010_planar: the per-row warp exists for the fisheye -> pinhole modes (0, 1, 5) only");
    if (!(pitch_y < (1u << 24) && pitch_uv < (1u << 24) && (uint64_t)pitch_y * sh < (1ull << 32)))
        return fail(VSTAB_ERR_INVALID, "vstab_warp_p010_planar: source pitch too large");
    WarpArgs wa;
    wa.y = static_cast<const uint8_t *>(y), wa.uv = static_cast<const uint8_t *>(uv), wa.dst = static_cast<uint8_t *>(dst_y), wa.dst_uv = static_cast<uint8_t *>(dst_uv);
    wa.pitch_y = pitch_y, wa.pitch_uv = pitch_uv, wa.pitch_dst = pitch_dst_y, wa.pitch_dst_uv = pitch_dst_uv;
    wa.sw = sw, wa.sh = sh, wa.dw = dw, wa.dh = dh;
    wa.p = {params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7],
            {params[8], params[9], params[10], params[11], params[12], params[13], params[14], params[15], params[16]}};
    const bool src16 = reinterpret_cast<uintptr_t>(y) % 16 == 0 && reinterpret_cast<uintptr_t>(uv) % 16 == 0 && pitch_y % 16 == 0 && pitch_uv % 16 == 0;
    const bool dst16 = reinterpret_cast<uintptr_t>(dst_y) % 16 == 0 && reinterpret_cast<uintptr_t>(dst_uv) % 16 == 0 && pitch_dst_y % 16 == 0 && pitch_dst_uv % 16 == 0;
    return launch_warp_planar(wa, params, map_mode, 10, blend, src16, dst16, rot_bottom, static_cast<hipStream_t>(stream));
}

extern "C" vstab_status vstab_cvt_bgr16_p010(const void *src_bgr16, size_t pitch_src, int width, int height, void *dst_y, size_t pitch_y, void *dst_uv,
                                             size_t pitch_uv, void *stream) {
    if (!src_bgr16 || !dst_y || !dst_uv) return fail(VSTAB_ERR_INVALID, "vstab_cvt_bgr16_p010: null pointer");
    if (width <= 0 || height <= 0 || width > 32767 || height > 32767) return fail(VSTAB_ERR_INVALID, "vstab_cvt_bgr16_p010: sizes must be in [1, 32767]");
    if (pitch_src < (size_t)width * 6 || pitch_src % 2 || pitch_y < (size_t)width * 2 || pitch_y % 2 || pitch_uv < (size_t)((width + 1) / 2) * 4 || pitch_uv % 4 ||
        reinterpret_cast<uintptr_t>(src_bgr16) % 2 || reinterpret_cast<uintptr_t>(dst_y) % 2 || reinterpret_cast<uintptr_t>(dst_uv) % 4)
        return fail(VSTAB_ERR_INVALID, "vstab_cvt_bgr16_p010: bad pitch or alignment (16-bit samples; chroma pairs 4-byte aligned)");
    hipLaunchKernelGGL(k_cvt_bgr10_p010, dim3(div_up(div_up(width, 2), 64), div_up(height, 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint8_t *>(src_bgr16), pitch_src, width, height, static_cast<uint8_t *>(dst_y), pitch_y, static_cast<uint8_t *>(dst_uv), pitch_uv);
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}

extern "C" vstab_status vstab_warp_p010(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int sw, int sh, const float params[17],
                                        const float *rot_bottom, int map_mode, int blend, void *dst, size_t pitch_dst, int dw, int dh, void *stream) {
    if (!y || !uv || !dst || !params) return fail(VSTAB_ERR_INVALID, "vstab_warp_p010: null pointer");
    if (sw < 4 || sh < 2 || (sw & 1) || (sh & 1) || dw <= 0 || dh <= 0 || sw > 32767 || sh > 32767 || dw > 32767 || dh > 32767)
        return fail(VSTAB_ERR_INVALID, "vstab_warp_p010: sizes must be in [1, 32767], source even and at least 4 x 2");
    if (pitch_y < (size_t)sw * 2 || pitch_uv < (size_t)sw * 2 || pitch_dst < (size_t)dw * 6 || pitch_y % 2 || pitch_uv % 4 || pitch_dst % 2 ||
        reinterpret_cast<uintptr_t>(y) % 2 || reinterpret_cast<uintptr_t>(uv) % 4 || reinterpret_cast<uintptr_t>(dst) % 2)
        return fail(VSTAB_ERR_INVALID, "vstab_warp_p010: bad pitch or alignment (16-bit samples; chroma pairs 4-byte aligned)");
    if (map_mode < VSTAB_MAP_CREATEMAP_CL || map_mode > VSTAB_MAP_CREATEMAP_CL_OPENCL) return fail(VSTAB_ERR_INVALID, "vstab_warp_p010: unknown map mode");
    if (blend != VSTAB_BLEND_EXACT && blend != VSTAB_BLEND_FP16) return fail(VSTAB_ERR_INVALID, "vstab_warp_p010: unknown blend");
    // The LDS-tiled kernel (the 8-bit hot kernel's structure with a 10:10:10 LDS pixel) serves the fisheye -> pinhole maps when
    // the planes allow its 16-byte staging loads; everything else takes the direct-gather kernel below.  Same results.
    const bool tiled_ok = (map_mode == VSTAB_MAP_CREATEMAP_CL || map_mode == VSTAB_MAP_FISH_TO_RECT || map_mode == VSTAB_MAP_CREATEMAP_CL_OPENCL) && sw >= 8 &&
                          reinterpret_cast<uintptr_t>(y) % 16 == 0 &&
                          reinterpret_cast<uintptr_t>(uv) % 16 == 0 && pitch_y % 16 == 0 && pitch_uv % 16 == 0 && pitch_y < (1u << 24) &&
                          pitch_uv < (1u << 24) && (uint64_t)pitch_y * sh < (1ull << 32);
    static const bool force_direct = getenv("VSTAB_P010_DIRECT") != nullptr;  // development: the direct-gather kernel for every call
    if (tiled_ok && !force_direct) {
        WarpArgs wa;
        wa.y = static_cast<const uint8_t *>(y), wa.uv = static_cast<const uint8_t *>(uv), wa.dst = static_cast<uint8_t *>(dst), wa.dst_uv = nullptr;
        wa.pitch_y = pitch_y, wa.pitch_uv = pitch_uv, wa.pitch_dst = pitch_dst, wa.pitch_dst_uv = 0;
        wa.sw = sw, wa.sh = sh, wa.dw = dw, wa.dh = dh;
        wa.p = {params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7],
                {params[8], params[9], params[10], params[11], params[12], params[13], params[14], params[15], params[16]}};
        return launch_warp_fused10(wa, params, map_mode, blend, rot_bottom, false, true, static_cast<hipStream_t>(stream));
    }
    P010Args a;
    a.y = static_cast<const uint8_t *>(y), a.uv = static_cast<const uint8_t *>(uv), a.dst = static_cast<uint16_t *>(dst);
    a.pitch_y = pitch_y, a.pitch_uv = pitch_uv, a.pitch_dst = pitch_dst;
    a.sw = sw, a.sh = sh, a.dw = dw, a.dh = dh;
    a.p = {params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7],
           {params[8], params[9], params[10], params[11], params[12], params[13], params[14], params[15], params[16]}};
    a.rs = rot_bottom != nullptr;
    for (int k = 0; k < 9; k++) a.rs_d[k] = rot_bottom ? rot_bottom[k] - params[8 + k] : 0.0f;
    a.rs_den = (float)(dh > 1 ? dh - 1 : 1);
    const dim3 grid(div_up(dw, 64), div_up(dh, 8));  // 64 columns x 4 pairs of rows per workgroup
    hipStream_t s = static_cast<hipStream_t>(stream);
    const LaunchEvents ev = take_launch_events();  // a profiling caller's pair: the kernel's own start / end stamps
#define VSTAB_LAUNCH_B(M, B)                                                                                   \
    do {                                                                                                       \
        if (ev.start) hipExtLaunchKernelGGL((k_warp_p010<M, B>), grid, dim3(256), 0, s, ev.start, ev.stop, 0, a); \
        else hipLaunchKernelGGL((k_warp_p010<M, B>), grid, dim3(256), 0, s, a);                                  \
    } while (0)
#define VSTAB_LAUNCH(M)                                              \
    if (blend == VSTAB_BLEND_FP16) VSTAB_LAUNCH_B(M, VSTAB_BLEND_FP16); \
    else VSTAB_LAUNCH_B(M, VSTAB_BLEND_EXACT)
    switch (map_mode) {
        case VSTAB_MAP_CREATEMAP_CL: VSTAB_LAUNCH(MAP_CREATEMAP_CL); break;
        case VSTAB_MAP_FISH_TO_RECT: VSTAB_LAUNCH(MAP_FISH_TO_RECT); break;
        case VSTAB_MAP_FISH_TO_FISH: VSTAB_LAUNCH(MAP_FISH_TO_FISH); break;
        case VSTAB_MAP_RECT_TO_RECT: VSTAB_LAUNCH(MAP_RECT_TO_RECT); break;
        case VSTAB_MAP_CREATEMAP_CL_OPENCL: VSTAB_LAUNCH(MAP_CREATEMAP_CL_OPENCL); break;
        default: VSTAB_LAUNCH(MAP_RECT_TO_FISH); break;
    }
#undef VSTAB_LAUNCH
#undef VSTAB_LAUNCH_B
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}
