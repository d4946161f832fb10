// vstab_warp_tile.hpp -- the phases the LDS-tiled warp kernels share (vstab_warp_fused.hip: NV12 / P010 -> BGR; vstab_warp_planar.hip:
// plane-wise NV12 -> NV12 / P010 -> P010): coordinate quantisation constants, the perimeter probe that finds a tile's source box,
// the exact map of a thread's row pairs in lock-step (IEEE and the reference kernel's arithmetic), and the XCD band / tile schedule.
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>

#include "vstab_device.hpp"
#include "vstab_internal.hpp"
#include "vstab_warp_args.hpp"

namespace vstab {

// cv::remap rounds 32 * map half to even (cvRound).  Adding 1.5 * 2^23 does that rounding in the float adder: for
// |a| < 2^22 the low mantissa bits of a + QMAGIC hold rint(a) (two's complement), i.e. bits - QMAGIC_BITS == rint(a).
// Every other input (|a| >= 2^22, +-inf, NaN) yields an integer >= 2^22 in magnitude, which lands far outside any
// source <= 32767 wide -- the same "outside" cv::remap reaches through cvRound -> INT_MIN.  QMAGIC_BITS is a multiple
// of 32, so (bits >> 5) - (QMAGIC_BITS >> 5) is the tap column and bits & 31 the fraction.
// Register budget of the kernel as waves per SIMD it must leave room for: 7 -> at most 72 registers (see k_warp_fused for the
// kernels that get 80).  The kernel itself is held to 4 waves per SIMD by its LDS; what it leaves free is what the tracker and
// pyramid kernels beside it run in.
#ifndef VSTAB_WARP_WAVES
#define VSTAB_WARP_WAVES 7
#endif
// Issue priority of the kernel's waves (the probe wave runs at 3).  The warp and the tracker (k_lk_track, VSTAB_LK_PRIO) run at 1, the pyramid
// and detector kernels at the default 0: beside a saturating warp those two take a fifth of its rate while they run (every warp launch's
// duration against what ran beside it: profiles/r04_warp_overlap_regression.txt), and they have slack.  The warp ABOVE the tracker loses:
// the tracker's single iterating wave then waits for issue slots and its chain becomes the limiter (-3 % at 4K, -9 % at 1080p).
#ifndef VSTAB_WARP_PRIO
#define VSTAB_WARP_PRIO 1
#endif
#ifndef VSTAB_MAP_GROUP
#define VSTAB_MAP_GROUP 2
#endif
#ifndef VSTAB_TAP_GROUP
#define VSTAB_TAP_GROUP 4
#endif
constexpr int MAP_GROUP = VSTAB_MAP_GROUP;  // row pairs whose exact-map chains advance in lock-step
constexpr int TAP_GROUP = VSTAB_TAP_GROUP;  // output rows whose LDS tap reads are issued before the first blend
constexpr float QMAGIC = 12582912.0f;
constexpr int QMAGIC_BITS = 0x4B400000;

typedef short short2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) uint32_t LdsWord;  // a dword in LDS, addressed by its 32-bit LDS address
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2 LdsPair;     // an 8-byte pixel (three halves) there

__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b)));
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b)));
}

// wave-wide reduction of packed (x, y) int16 pairs: four row_shr steps inside each 16-lane row, then row_bcast15 and
// row_bcast31 carry the row totals across the rows; lane 63 ends up with the total of all 64 lanes.
template <bool MAX>
__device__ __forceinline__ uint32_t wave_reduce_pk_i16(uint32_t v) {
    constexpr uint32_t ident = MAX ? 0x80008000u : 0x7fff7fffu;
#define VSTAB_STEP(ctrl, rmask)                                                                              \
    {                                                                                                        \
        const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)ident, (int)v, ctrl, rmask, 0xf, false); \
        v = MAX ? pk_max_i16(v, t) : pk_min_i16(v, t);                                                       \
    }
    VSTAB_STEP(0x111, 0xf) VSTAB_STEP(0x112, 0xf) VSTAB_STEP(0x114, 0xf) VSTAB_STEP(0x118, 0xf)
    VSTAB_STEP(0x142, 0xa) VSTAB_STEP(0x143, 0xc)
#undef VSTAB_STEP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// One pixel sampled straight from global memory (per-tap zeroing): the rare path.  sx, sy = rint(32 * map).
__device__ __forceinline__ uint32_t gather_pixel_far(const WarpArgs &a, int sx, int sy) {
    const int X = sx >> 5, Y = sy >> 5;
    if (!(X < a.sw && X + 1 >= 0 && Y < a.sh && Y + 1 >= 0)) return 0;
    return gather_pixel(a, sx, sy);
}

// Trips of the staging loop: STAGE_MAX * 256 blocks of 8 x 2 source pixels must cover the largest box the tile's LDS
// budget admits (40 KB for kernels whose tall tiles have 32 rows, 24 KB for 16; half-height tiles have smaller boxes).
template <int RWB, int RW>
struct StageTrips {
    static constexpr int value = RW == 8 ? 3 : RW == 4 ? 2 : (RWB == 8 ? 2 : 1);
};

// The source bounding box of the output tile at (x0, y0), 64 x TH pixels, from the map on 64 perimeter pixels of the
// tile (one per lane; a continuous map attains its coordinate extremes on the perimeter).  Lane 0 writes
// {bx0, by0, wb, hb, use_lds} to hdr.  Run by one wave.
// PLANAR (vstab_warp_planar.hip): the box also has to hold the taps of the chroma plane, whose positions are the luma positions of the
// even pixels halved and rounded again -- up to three luma pixels further right / down and one further left / up than the luma taps --
// and it is staged in blocks of BLOCK_W (a power of two) source pixels, with a border of black blocks up to two pixels outside the source.
template <int TH, int STAGE_MAX, int MODE, bool CACHED, bool PLANAR = false, int BLOCK_W = 8>
__device__ __forceinline__ void probe_tile(const FusedArgs &ta, int x0, int y0, int lane, float rfx, float rfy, uint32_t *hdr) {
    constexpr bool RS = map_mode_is_rs(MODE);    // per-row rotation
    constexpr int BASE = map_mode_base(MODE);    // the projection pair and its arithmetic
    const WarpArgs &a = ta.w;
    int px, py;  // tile-local perimeter point of this lane: 16 along the top, 16 along the bottom, 16 per side
    const int l16 = lane & 15, side = (l16 * (TH - 1) + 7) / 15;
    if (lane < 16) px = 4 * l16, py = 0;
    else if (lane < 32) px = 4 * l16 + 3, py = TH - 1;
    else if (lane < 48) px = 0, py = side;
    else px = 63, py = side;
    px = min(px, a.dw - 1 - x0), py = min(py, a.dh - 1 - y0);
    int qx, qy;
    if constexpr (CACHED) {
        const int2 q = ta.qmap[(size_t)(y0 + py) * ta.qpitch + (x0 + px)];
        qx = q.x, qy = q.y;
    } else {
        float ax, ay;
        if constexpr (BASE == MAP_CREATEMAP_CL || BASE == MAP_FISH_TO_RECT || BASE == MAP_CREATEMAP_CL_OPENCL) {
            // The box needs the map to a fraction of a pixel only (it has a pixel of margin and never decides a result),
            // so the probe uses the approximate reciprocal / rsqrt instructions and fused operations: a third of the
            // dependent chain of the exact evaluation, on the one wave the other three are waiting for.
            const float vx = ((float)(x0 + px) - a.p.ocx) * rfx, vy = ((float)(y0 + py) - a.p.ocy) * rfy;
            float m[9];
#pragma unroll
            for (int k = 0; k < 9; k++) m[k] = a.p.r[k];
            if constexpr (RS) {  // the matrix of this lane's row, to probe accuracy
                const float t = (float)(y0 + py) * __builtin_amdgcn_rcpf(ta.rs_den);
#pragma unroll
                for (int k = 0; k < 9; k++) m[k] = __builtin_fmaf(t, ta.rs_d[k], a.p.r[k]);
            }
            const float wx = __builtin_fmaf(m[0], vx, __builtin_fmaf(m[1], vy, m[2]));
            const float wy = __builtin_fmaf(m[3], vx, __builtin_fmaf(m[4], vy, m[5]));
            const float wz = __builtin_fmaf(m[6], vx, __builtin_fmaf(m[7], vy, m[8]));
            const float rz = __builtin_amdgcn_rcpf(wz), ux = wx * rz, uy = wy * rz;
            const float q = __builtin_fmaf(ux, ux, uy * uy), rs = __builtin_amdgcn_rsqf(q), rad = q * rs;
            const bool inv = rad > 1.0f;
            const float t = inv ? rs : rad, s2 = t * t;
            float g = 0.0028423243202269077f;
            g = __builtin_fmaf(g, s2, -0.016053270548582077f);
            g = __builtin_fmaf(g, s2, 0.04269874095916748f);
            g = __builtin_fmaf(g, s2, -0.07508683204650879f);
            g = __builtin_fmaf(g, s2, 0.1064559817314148f);
            g = __builtin_fmaf(g, s2, -0.14205896854400635f);
            g = __builtin_fmaf(g, s2, 0.19993145763874054f);
            g = __builtin_fmaf(g, s2, -0.33333125710487366f);
            float at = __builtin_fmaf(t * s2, g, t);
            at = inv ? 1.57079637050628662109375f - at : at;
            const float k = at * rs;  // atan(rad) / rad; NaN on the axis (q == 0) only widens the box
            ax = __builtin_fmaf(ux * k, ta.p32.ifx32, ta.p32.icx32), ay = __builtin_fmaf(uy * k, ta.p32.ify32, ta.p32.icy32);
            if (BASE == MAP_FISH_TO_RECT && !(wz > 0.0f)) ax = ay = __builtin_nanf("");
        } else {
            const float vx = div_with_rcp((float)(x0 + px) - a.p.ocx, a.p.ofx, rfx);
            const float vy = div_with_rcp((float)(y0 + py) - a.p.ocy, a.p.ofy, rfy);
            const ColTerm ct = {a.p.r[0] * vx, a.p.r[3] * vx, a.p.r[6] * vx};
            const RowTerm rt = {a.p.r[1] * vy, a.p.r[4] * vy, a.p.r[7] * vy};
            map_pixel_ex<BASE>(ta.p32, a.p, ct, rt, vx, vy, ax, ay);
        }
        qx = __float_as_int(ax + QMAGIC) - QMAGIC_BITS, qy = __float_as_int(ay + QMAGIC) - QMAGIC_BITS;
    }
    // clamp to one step outside the source: pixels that map outside pull the box to the nearest edge only
    int Xc, Yc;
    asm("v_med3_i32 %0, %1, -1, %2" : "=v"(Xc) : "v"(qx >> 5), "s"(a.sw));
    asm("v_med3_i32 %0, %1, -1, %2" : "=v"(Yc) : "v"(qy >> 5), "s"(a.sh));
    const uint32_t pk = ((uint32_t)Xc & 0xffffu) | ((uint32_t)Yc << 16);
    const uint32_t mn = wave_reduce_pk_i16<false>(pk), mx = wave_reduce_pk_i16<true>(pk);
    const int mnx = (int)(short)(mn & 0xffffu), mny = (int)mn >> 16, mxx = (int)(short)(mx & 0xffffu), mxy = (int)mx >> 16;
    // Columns the taps of the tile can touch: [min - 1, max + 2] with one pixel of margin (the perimeter is sampled
    // every 2 to 4 pixels), cut to [-1, sw]: the staged box carries a border of zero pixels where it leaves the source
    // (cv::remap's BORDER_CONSTANT), so pixels whose footprint straddles the source edge are sampled like all others.
    constexpr int LO_M = PLANAR ? 2 : 1, HI_M = PLANAR ? 5 : 2, CAP = PLANAR ? 1 : 0, BM = BLOCK_W - 1;
    const int lox = max(mnx - LO_M, -LO_M), hix = min(mxx + HI_M, a.sw + CAP), loy = max(mny - LO_M, -LO_M), hiy = min(mxy + HI_M, a.sh + CAP);
    const int bx0 = lox & ~BM, by0 = loy & ~1;  // -BLOCK_W / -2 when the box starts left of / above the source
    const int wb = (hix + 1 - bx0 + BM) & ~BM, hb = (hiy + 1 - by0 + 1) & ~1;
    const bool have = mnx < a.sw && mxx >= -1 && mny < a.sh && mxy >= -1;  // else every pixel of the tile is outside
    // staged as whole BLOCK_W x 2 blocks with aligned vector loads: unaligned planes, and boxes that reach the last,
    // partial block column of a source whose width is not a multiple of BLOCK_W, are sampled straight from global memory
    const bool stageable = have && a.sw >= BLOCK_W && ta.src_vec_ok && ((a.sw & BM) == 0 || bx0 + wb <= (a.sw & ~BM));
#ifdef VSTAB_DEV
    const bool fits = (wb + ta.lds_pad) * hb <= ta.lds_capacity_px && (wb >> 3) * (hb >> 1) <= STAGE_MAX * 256;  // experiment: padded LDS rows
#else
    const bool fits = wb * hb <= ta.lds_capacity_px && (wb / BLOCK_W) * (hb >> 1) <= STAGE_MAX * 256;
#endif
    if (lane == 0) {
        *reinterpret_cast<uint4 *>(hdr) = make_uint4((uint32_t)bx0, (uint32_t)by0, (uint32_t)wb, (uint32_t)hb);
        hdr[4] = stageable ? (fits ? 1u : 2u) : 0u;  // 2: the box is over the LDS budget -- a tall tile is then done as two half-height tiles
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The exact map of NP row pairs of one column (lane), IEEE mode (MAP_CREATEMAP_CL / MAP_FISH_TO_RECT): the operation
// sequence of map_pixel32_x2 (vstab_device.hpp) for every pair, written step by step ACROSS the pairs, so that NP
// independent packed instructions stand between an instruction and the one that needs its result (the compiler
// otherwise emits the chains one after the other, separated by the s_nop a dependent packed-fp32 instruction needs).
// ---------------------------------------------------------------------------------------------------------------------
#define VSTAB_EACH _Pragma("unroll") for (int c = 0; c < NP; c++)
template <int NP, bool FISH_TO_RECT>
__device__ __forceinline__ void map_pairs_ieee(float icx32, float icy32, float ifx32, float ify32, const f32x2 (&wx)[NP], const f32x2 (&wy)[NP],
                                               const f32x2 (&wz)[NP], f32x2 (&ax)[NP], f32x2 (&ay)[NP]) {
    f32x2 rz[NP], e[NP], px[NP], py[NP], ex[NP], ey[NP], q[NP], y[NP], g[NP], h[NP], d[NP], rad[NP], rr[NP], t[NP], s[NP], at[NP], k[NP];
    i32x2 inv[NP];
    const f32x2 one = splat2(1.0f), half = splat2(0.5f);
    // rz = rcp_refined2(wz)
    VSTAB_EACH rz[c] = (f32x2){__builtin_amdgcn_rcpf(wz[c].x), __builtin_amdgcn_rcpf(wz[c].y)};
    VSTAB_EACH e[c] = fma2(-wz[c], rz[c], one);
    VSTAB_EACH rz[c] = fma2(e[c], rz[c], rz[c]);
    // px = div_with_rcp2(wx, wz, rz), py = div_with_rcp2(wy, wz, rz)
    VSTAB_EACH px[c] = wx[c] * rz[c], py[c] = wy[c] * rz[c];
    VSTAB_EACH ex[c] = fma2(-wz[c], px[c], wx[c]), ey[c] = fma2(-wz[c], py[c], wy[c]);
    VSTAB_EACH px[c] = fma2(ex[c], rz[c], px[c]), py[c] = fma2(ey[c], rz[c], py[c]);
    VSTAB_EACH ex[c] = fma2(-wz[c], px[c], wx[c]), ey[c] = fma2(-wz[c], py[c], wy[c]);
    VSTAB_EACH px[c] = fma2(ex[c], rz[c], px[c]), py[c] = fma2(ey[c], rz[c], py[c]);
    // q = px * px + py * py ; rad = sqrt_rn2(q)
    VSTAB_EACH q[c] = px[c] * px[c] + py[c] * py[c];
    VSTAB_EACH y[c] = (f32x2){__builtin_amdgcn_rsqf(q[c].x), __builtin_amdgcn_rsqf(q[c].y)};
    VSTAB_EACH g[c] = q[c] * y[c], h[c] = half * y[c];
    VSTAB_EACH e[c] = fma2(-h[c], g[c], half);
    VSTAB_EACH h[c] = fma2(h[c], e[c], h[c]), g[c] = fma2(g[c], e[c], g[c]);
    VSTAB_EACH d[c] = fma2(-g[c], g[c], q[c]);
    VSTAB_EACH rad[c] = fma2(d[c], h[c], g[c]);
    // rr = rcp_refined2(rad)
    VSTAB_EACH rr[c] = (f32x2){__builtin_amdgcn_rcpf(rad[c].x), __builtin_amdgcn_rcpf(rad[c].y)};
    VSTAB_EACH e[c] = fma2(-rad[c], rr[c], one);
    VSTAB_EACH rr[c] = fma2(e[c], rr[c], rr[c]);
    // t = inv ? div_with_rcp2(1, rad, rr) : rad
    VSTAB_EACH inv[c] = rad[c] > one, t[c] = one * rr[c];
    VSTAB_EACH e[c] = fma2(-rad[c], t[c], one);
    VSTAB_EACH t[c] = fma2(e[c], rr[c], t[c]);
    VSTAB_EACH e[c] = fma2(-rad[c], t[c], one);
    VSTAB_EACH t[c] = fma2(e[c], rr[c], t[c]);
    VSTAB_EACH t[c] = inv[c] ? t[c] : rad[c];
    VSTAB_EACH s[c] = t[c] * t[c];
    // atan_pos polynomial
    VSTAB_EACH g[c] = fma2(splat2(0.0028423243202269077f), s[c], splat2(-0.016053270548582077f));
    VSTAB_EACH g[c] = fma2(g[c], s[c], splat2(0.04269874095916748f));
    VSTAB_EACH g[c] = fma2(g[c], s[c], splat2(-0.07508683204650879f));
    VSTAB_EACH g[c] = fma2(g[c], s[c], splat2(0.1064559817314148f));
    VSTAB_EACH g[c] = fma2(g[c], s[c], splat2(-0.14205896854400635f));
    VSTAB_EACH g[c] = fma2(g[c], s[c], splat2(0.19993145763874054f));
    VSTAB_EACH g[c] = fma2(g[c], s[c], splat2(-0.33333125710487366f));
    VSTAB_EACH at[c] = fma2(t[c] * s[c], g[c], t[c]);
    VSTAB_EACH at[c] = inv[c] ? (splat2(1.57079637050628662109375f) - at[c]) + splat2(-4.37113900018624283e-8f) : at[c];
    // k = div_with_rcp2(at, rad, rr)
    VSTAB_EACH k[c] = at[c] * rr[c];
    VSTAB_EACH e[c] = fma2(-rad[c], k[c], at[c]);
    VSTAB_EACH k[c] = fma2(e[c], rr[c], k[c]);
    VSTAB_EACH e[c] = fma2(-rad[c], k[c], at[c]);
    VSTAB_EACH k[c] = fma2(e[c], rr[c], k[c]);
    if constexpr (FISH_TO_RECT) {
        VSTAB_EACH k[c] = (q[c] == splat2(0.0f)) ? one : k[c];
    }
    VSTAB_EACH ax[c] = splat2(icx32) + (px[c] * k[c]) * splat2(ifx32), ay[c] = splat2(icy32) + (py[c] * k[c]) * splat2(ify32);
    if constexpr (FISH_TO_RECT) {
        VSTAB_EACH {
            const i32x2 ok = wz[c] > splat2(0.0f);
            ax[c] = ok ? ax[c] : splat2(__builtin_nanf("")), ay[c] = ok ? ay[c] : splat2(__builtin_nanf(""));
        }
    }
}

// The same for MAP_CREATEMAP_CL_OPENCL: map_pixel_ocl_fast (vstab_device.hpp) on NP row pairs in lock-step.  qmin / qmax
// collect the range of q (its bits as unsigned) over all the caller's pixels: when some intermediate may have left the
// normal range the caller re-evaluates the wave's pixels with the code object's literal instruction stream.
// q in [2^-80, 2^80] is the whole test (map_q_irregular): a rotated ray has a component >= 1/2, so a reciprocal, quotient
// or square that over- or underflows shows up as q = inf / NaN / tiny.
__device__ __forceinline__ bool map_q_irregular(uint32_t qmin, uint32_t qmax) { return qmin < 0x17800000u /* 2^-80 */ || qmax > 0x67800000u /* 2^80 */; }
template <int NP>
__device__ __forceinline__ void map_pairs_ocl(float icx32, float icy32, float ifx32, float ify32, const f32x2 (&wx)[NP], const f32x2 (&wy)[NP],
                                              const f32x2 (&wz)[NP], f32x2 (&ax)[NP], f32x2 (&ay)[NP], uint32_t &qmin, uint32_t &qmax) {
    f32x2 rz[NP], px[NP], py[NP], q[NP], rad[NP], rr[NP], t[NP], s[NP], p[NP], r[NP], k[NP];
    i32x2 inv[NP];
    VSTAB_EACH rz[c] = (f32x2){__builtin_amdgcn_rcpf(wz[c].x), __builtin_amdgcn_rcpf(wz[c].y)};
    VSTAB_EACH px[c] = wx[c] * rz[c], py[c] = wy[c] * rz[c];
    VSTAB_EACH q[c] = fma2(py[c], py[c], px[c] * px[c]);
    VSTAB_EACH rad[c] = (f32x2){__builtin_amdgcn_sqrtf(q[c].x), __builtin_amdgcn_sqrtf(q[c].y)};
    VSTAB_EACH {
        qmin = min(min(qmin, __float_as_uint(q[c].x)), __float_as_uint(q[c].y));
        qmax = max(max(qmax, __float_as_uint(q[c].x)), __float_as_uint(q[c].y));
    }
    VSTAB_EACH rr[c] = (f32x2){__builtin_amdgcn_rcpf(rad[c].x), __builtin_amdgcn_rcpf(rad[c].y)};
    VSTAB_EACH inv[c] = rad[c] > splat2(1.0f), t[c] = inv[c] ? rr[c] : rad[c];
    VSTAB_EACH s[c] = t[c] * t[c];
    VSTAB_EACH p[c] = fma2(splat2(f32_bits(0x3b2d2a58u)), s[c], splat2(f32_bits(0xbc7a590cu)));
    VSTAB_EACH p[c] = fma2(s[c], p[c], splat2(f32_bits(0x3d29fb3fu)));
    VSTAB_EACH p[c] = fma2(s[c], p[c], splat2(f32_bits(0xbd97d4d7u)));
    VSTAB_EACH p[c] = fma2(s[c], p[c], splat2(f32_bits(0x3dd931b2u)));
    VSTAB_EACH p[c] = fma2(s[c], p[c], splat2(f32_bits(0xbe1160e6u)));
    VSTAB_EACH p[c] = fma2(s[c], p[c], splat2(f32_bits(0x3e4cb8bfu)));
    VSTAB_EACH p[c] = fma2(s[c], p[c], splat2(f32_bits(0xbeaaaa62u)));
    VSTAB_EACH r[c] = fma2(t[c], s[c] * p[c], t[c]);
    VSTAB_EACH r[c] = inv[c] ? splat2(f32_bits(0x3fc90fdbu)) - r[c] : r[c];
    VSTAB_EACH k[c] = r[c] * rr[c];
    VSTAB_EACH ax[c] = fma2(splat2(ifx32), px[c] * k[c], splat2(icx32)), ay[c] = fma2(splat2(ify32), py[c] * k[c], splat2(icy32));
}
#undef VSTAB_EACH

// ---------------------------------------------------------------------------------------------------------------------
// The map phase of a tile: the exact map of the thread's RW pixels (lane = column x, rows y_first .. y_first + RW - 1), quantised as
// cv::remap does: qxb / qyb = the bits of (32 * map + QMAGIC) (see QMAGIC above; CACHED: the quantised map's integers themselves).
// Columns right of the image and rows below it are evaluated as the last column / row (never stored; they stay inside the box).
// CHROMA (plane-wise warp): also the quantised position in the chroma plane for the EVEN rows of the thread, meaningful in lanes
// with an even column: qcx / qcy = the bits of (32 * (map / 2) + QMAGIC) -- the halving is exact, the rounding is cv::remap's again.
// ---------------------------------------------------------------------------------------------------------------------
// FOLD (plane-wise warp): the rounding constants carry the origin of the staged box -- (32 * map - 32 * origin) + QMAGIC rounds to the
// same integer minus 32 * origin, which is an integer -- so the registers hold box-relative positions and a tap address is two bit-field
// extracts and a multiply-add.  qm = {QMAGIC - 32 * bx0, QMAGIC - 32 * by0, QMAGIC - 32 * (bx0 / 2), QMAGIC - 32 * (by0 / 2)}.
__device__ __forceinline__ void chroma_quantise(float ax32, float ay32, float mx, float my, int &qcx, int &qcy) {
    qcx = __float_as_int(ax32 * 0.5f + mx), qcy = __float_as_int(ay32 * 0.5f + my);
}
template <int RW, int MODE, bool CACHED, bool CHROMA, bool FOLD = false>
__device__ __forceinline__ void map_phase(const FusedArgs &ta, const int x, const int y0, const int wave, const int lane, const float rfx, const float rfy,
                                          int (&qxb)[RW], int (&qyb)[RW], int (&qcx)[RW / 2], int (&qcy)[RW / 2], const float (&qm)[4]) {
    static_assert(!(CACHED && CHROMA), "the quantised map holds no chroma positions");
    const float QMX = FOLD ? qm[0] : QMAGIC, QMY = FOLD ? qm[1] : QMAGIC, QCX = FOLD ? qm[2] : QMAGIC, QCY = FOLD ? qm[3] : QMAGIC;
    constexpr bool RS = map_mode_is_rs(MODE);
    constexpr int BASE = map_mode_base(MODE);
    const WarpArgs &a = ta.w;
    if constexpr (CACHED) {
#pragma unroll
        for (int j = 0; j < RW; j++) {
            const int y = min(y0 + wave * RW + j, a.dh - 1);
            const int2 q = ta.qmap[(size_t)y * ta.qpitch + min(x, a.dw - 1)];
            qxb[j] = q.x, qyb[j] = q.y;
        }
    } else
#ifdef VSTAB_DEV
    if (ta.ablate & 1) {  // timing only: a linear map in place of the exact one
#pragma unroll
        for (int j = 0; j < RW; j++) {
            const float fx = (float)x * (32.0f * (float)a.sw / (float)a.dw), fy = (float)(y0 + wave * RW + j) * (32.0f * (float)a.sh / (float)a.dh);
            qxb[j] = __float_as_int(fx + QMX), qyb[j] = __float_as_int(fy + QMY);
        }
    } else
#endif
    {
        float vx = norm_coord<BASE>((float)min(x, a.dw - 1) - a.p.ocx, a.p.ofx, rfx);
        // the compiler would sink each map evaluation to its use behind the barrier; these two statements pin the map
        // phase between the loads above (memory clobber) and the conversion below (the coordinates pass through)
        asm volatile("" : "+v"(vx) : : "memory");
        const ColTerm ct = {a.p.r[0] * vx, a.p.r[3] * vx, a.p.r[6] * vx};
        // row terms: lane l < RW evaluates row l of this wave once; every lane then reads them from that lane
        const int y_l = min(y0 + wave * RW + (lane & (RW - 1)), a.dh - 1);
        const float vy_l = norm_coord<BASE>((float)y_l - a.p.ocy, a.p.ofy, rfy);
        float m_l[9];  // the rotation of row y_l: the frame's, or interpolated towards the last row's (rolling shutter)
#pragma unroll
        for (int k = 0; k < 9; k++) m_l[k] = a.p.r[k];
        if constexpr (RS) {
            const float t = div_with_rcp((float)y_l, ta.rs_den, rcp_refined(ta.rs_den));
#pragma unroll
            for (int k = 0; k < 9; k++) m_l[k] = __builtin_fmaf(t, ta.rs_d[k], a.p.r[k]);
        }
        const float b0_l = m_l[1] * vy_l, b1_l = m_l[4] * vy_l, b2_l = m_l[7] * vy_l;
        const float icx32 = ta.p32.icx32, icy32 = ta.p32.icy32, ifx32 = ta.p32.ifx32, ify32 = ta.p32.ify32;
        auto bcast = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
        auto bcast2 = [&bcast](float v, int j) { return (f32x2){bcast(v, j), bcast(v, j + 1)}; };
        if constexpr (BASE == MAP_CREATEMAP_CL || BASE == MAP_FISH_TO_RECT) {
            // row pairs in lock-step groups of MAP_GROUP (more pairs in flight would push the kernel past 64 registers and
            // take wave slots away from the tracker and pyramid kernels that run beside it)
            constexpr int NP = RW / 2 < MAP_GROUP ? RW / 2 : MAP_GROUP;
#pragma unroll
            for (int g0 = 0; g0 < RW / 2; g0 += NP) {
                f32x2 wx[NP], wy[NP], wz[NP], ax[NP], ay[NP];
#pragma unroll
                for (int c = 0; c < NP; c++) {
                    const int j = 2 * (g0 + c);
                    if constexpr (RS) {  // every row has its own matrix: column products and third column per row
                        wx[c] = (bcast2(m_l[0], j) * splat2(vx) + bcast2(b0_l, j)) + bcast2(m_l[2], j);
                        wy[c] = (bcast2(m_l[3], j) * splat2(vx) + bcast2(b1_l, j)) + bcast2(m_l[5], j);
                        wz[c] = (bcast2(m_l[6], j) * splat2(vx) + bcast2(b2_l, j)) + bcast2(m_l[8], j);
                    } else {
                        wx[c] = (splat2(ct.a0) + bcast2(b0_l, j)) + splat2(a.p.r[2]);
                        wy[c] = (splat2(ct.a1) + bcast2(b1_l, j)) + splat2(a.p.r[5]);
                        wz[c] = (splat2(ct.a2) + bcast2(b2_l, j)) + splat2(a.p.r[8]);
                    }
                }
                map_pairs_ieee<NP, BASE == MAP_FISH_TO_RECT>(icx32, icy32, ifx32, ify32, wx, wy, wz, ax, ay);
#pragma unroll
                for (int c = 0; c < NP; c++) {
                    const int j = 2 * (g0 + c);
                    if constexpr (CHROMA) chroma_quantise(ax[c].x, ay[c].x, QCX, QCY, qcx[g0 + c], qcy[g0 + c]);
                    ax[c] += splat2(QMX), ay[c] += splat2(QMY);
                    qxb[j] = __float_as_int(ax[c].x), qxb[j + 1] = __float_as_int(ax[c].y);
                    qyb[j] = __float_as_int(ay[c].x), qyb[j + 1] = __float_as_int(ay[c].y);
                }
            }
        } else if constexpr (BASE == MAP_CREATEMAP_CL_OPENCL) {
            constexpr int NP = RW / 2 < MAP_GROUP ? RW / 2 : MAP_GROUP;
            uint32_t qmin = 0xffffffffu, qmax = 0u;  // one range test for all RW pixels (min3 / max3: half an instruction per pixel)
#pragma unroll
            for (int g0 = 0; g0 < RW / 2; g0 += NP) {
                f32x2 wx[NP], wy[NP], wz[NP], ax[NP], ay[NP];
#pragma unroll
                for (int c = 0; c < NP; c++) {
                    const int j = 2 * (g0 + c);
                    const f32x2 vy2 = bcast2(vy_l, j);
                    if constexpr (RS) {  // every row has its own matrix (the definition's fp32 interpolation), fed to createMap.cl's stream
                        wz[c] = fma2(bcast2(m_l[7], j), vy2, bcast2(m_l[6], j) * splat2(vx)) + bcast2(m_l[8], j);
                        wx[c] = fma2(bcast2(m_l[1], j), vy2, bcast2(m_l[0], j) * splat2(vx)) + bcast2(m_l[2], j);
                        wy[c] = fma2(bcast2(m_l[4], j), vy2, bcast2(m_l[3], j) * splat2(vx)) + bcast2(m_l[5], j);
                    } else {
                        wz[c] = fma2(splat2(a.p.r[7]), vy2, splat2(ct.a2)) + splat2(a.p.r[8]);
                        wx[c] = fma2(splat2(a.p.r[1]), vy2, splat2(ct.a0)) + splat2(a.p.r[2]);
                        wy[c] = fma2(splat2(a.p.r[4]), vy2, splat2(ct.a1)) + splat2(a.p.r[5]);
                    }
                }
                map_pairs_ocl<NP>(icx32, icy32, ifx32, ify32, wx, wy, wz, ax, ay, qmin, qmax);
#pragma unroll
                for (int c = 0; c < NP; c++) {
                    const int j = 2 * (g0 + c);
                    if constexpr (CHROMA) chroma_quantise(ax[c].x, ay[c].x, QCX, QCY, qcx[g0 + c], qcy[g0 + c]);
                    ax[c] += splat2(QMX), ay[c] += splat2(QMY);
                    qxb[j] = __float_as_int(ax[c].x), qxb[j + 1] = __float_as_int(ax[c].y);
                    qyb[j] = __float_as_int(ay[c].x), qyb[j + 1] = __float_as_int(ay[c].y);
                }
            }
            if (__builtin_amdgcn_ballot_w64(map_q_irregular(qmin, qmax))) {  // practically never: the code object's literal stream
#pragma unroll 1
                for (int j = 0; j < RW; j++) {
                    float fx, fy;
                    if constexpr (RS) {
                        float mr[9];
#pragma unroll
                        for (int k = 0; k < 9; k++) mr[k] = bcast(m_l[k], j);
                        map_pixel_ocl_literal(icx32, icy32, ifx32, ify32, mr, mr[0] * vx, mr[3] * vx, mr[6] * vx, bcast(vy_l, j), fx, fy);
                    } else {
                        map_pixel_ocl_literal(icx32, icy32, ifx32, ify32, a.p.r, ct.a0, ct.a1, ct.a2, bcast(vy_l, j), fx, fy);
                    }
                    const int ix = __float_as_int(fx + QMX), iy = __float_as_int(fy + QMY);
#pragma unroll
                    for (int k = 0; k < RW; k++) qxb[k] = j == k ? ix : qxb[k], qyb[k] = j == k ? iy : qyb[k];
                    if constexpr (CHROMA) {
                        int cx, cy;
                        chroma_quantise(fx, fy, QCX, QCY, cx, cy);
#pragma unroll
                        for (int k = 0; k < RW / 2; k++) qcx[k] = j == 2 * k ? cx : qcx[k], qcy[k] = j == 2 * k ? cy : qcy[k];
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < RW; j++) {
                const float vy = bcast(vy_l, j);
                const RowTerm rt = {bcast(b0_l, j), bcast(b1_l, j), bcast(b2_l, j)};
                float fx, fy;
                map_pixel_ex<BASE>(ta.p32, a.p, ct, rt, vx, vy, fx, fy);
                qxb[j] = __float_as_int(fx + QMX), qyb[j] = __float_as_int(fy + QMY);
                if constexpr (CHROMA) {
                    if (!(j & 1)) chroma_quantise(fx, fy, QCX, QCY, qcx[j / 2], qcy[j / 2]);
                }
            }
        }
    }
}

// Bands and tile heights of a launch (see k_warp_fused): the image's half-height tile rows dealt evenly to the 8 XCDs; inside
// a band the tall tiles come first, the last `tail_rounds` rounds of the XCD's workgroup slots (32 CUs x workgroups per CU)
// are made of half-height tiles.  Returns the grid size.
static inline unsigned tile_schedule(FusedArgs &ta, int rwb, int lds_kb, double tail_rounds) {
    const WarpArgs &a = ta.w;
    ta.lds_capacity_px = lds_kb * 1024 / 4 - 8;  // 8 dwords hold the tile header
    ta.tiles_x = (int)div_up(a.dw, 64);
    const int th = 4 * rwb, ts = th / 2;
    const int half_rows = (int)div_up(a.dh, ts);
    const int slots = 32 * std::max(1, std::min(8, (int)(160 / lds_kb)));
    int share = 0;
    for (int k = 0; k <= 8; k++) ta.band_y[k] = std::min(a.dh, (int)((long)k * half_rows / 8) * ts);
    ta.band_y[8] = a.dh;
    for (int k = 0; k < 8; k++) {
        const int rows = ta.band_y[k + 1] - ta.band_y[k];
        const int tall_rows_max = rows / th;  // whole tall tile rows that fit
        // half-height tiles for about tail_rounds * slots tall-tile equivalents at the end of the band
        const int tail_tall_rows = (int)std::min<long>(tall_rows_max, std::lround(tail_rounds * slots / ta.tiles_x));
        const int tall_rows = tall_rows_max - tail_tall_rows;
        ta.split_y[k] = ta.band_y[k] + tall_rows * th;
        const int n = tall_rows * ta.tiles_x + (int)div_up(ta.band_y[k + 1] - ta.split_y[k], ts) * ta.tiles_x;
        share = std::max(share, n);
    }
    return 8u * (unsigned)share;
}


}  // namespace vstab
