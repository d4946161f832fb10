// vstab_warp_fused.hip -- the fused undistort-remap kernel of the hot path (gfx950 / MI355X).
//
//   createMap.cl:13-50  ->  cv::remap quantisation (FrameSourceWarp.cpp:306-312)  ->  NV12 taps converted with the
//   cvtColor arithmetic (:401)  ->  fixed-point bilinear blend  ->  BGR8 (or NV12).
// Converting each tap and then blending is exactly what the reference's cvtColor-then-remap sequence computes, so the
// output is bit-identical to the un-fused operators.  Compiled with -ffp-contract=off.
//
// One workgroup (4 waves) per 64 x 4R output tile; lane = output column, a wave owns R consecutive output rows.
//   probe    wave 0 evaluates the map on 64 perimeter pixels of the tile and reduces them (DPP) to the tile's source
//            bounding box.  A continuous map attains its coordinate extremes on the perimeter.  One barrier.
//   load     all global loads of the thread's share of the box (8 x 2 blocks: two luma rows + one chroma row, 8-byte
//            loads) are issued at once ...
//   map      ... and the R exact map evaluations of the thread (hand-scheduled IEEE arithmetic, two rows at a time
//            through the packed-fp32 pipe) run underneath their latency.  Quantised coordinates stay in registers.
//   convert  the loaded blocks are converted ONCE per source pixel to BGRx with the cvtColor fixed-point arithmetic and
//            written to LDS (ds_write_b128); blocks outside the source become the zero border of BORDER_CONSTANT.
//            One barrier.
//   sample   four ds_read per pixel (lanes walk along a source row: at most 2-way bank conflicts), exact fixed-point
//            blend, rows transposed to 4-byte-per-lane stores with ds_bpermute.
// Bit-exactness never depends on the box: a pixel whose 2x2 footprint is not inside the staged box (degenerate
// rotations, a box larger than the LDS budget, unaligned planes) is sampled straight from global memory with per-tap
// zeroing (gather_pixel), which computes the same integers.  HBM traffic = the NV12 frame once + the output once.
#include <algorithm>
#include <climits>
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
constexpr float QMAGIC = 12582912.0f;
constexpr int QMAGIC_BITS = 0x4B400000;

typedef short short2v __attribute__((ext_vector_type(2)));

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

// The source bounding box of output tile (tile_x, tile_y), from the map on 64 perimeter pixels of the tile (one per lane;
// a continuous map attains its coordinate extremes on the perimeter).  Lane 0 writes {bx0, by0, wb, hb, use_lds, tile
// index} to hdr.  Run by one wave.
template <int R, int MODE, bool CACHED>
__device__ __forceinline__ void probe_tile(const FusedArgs &ta, int tile_x, int tile_y, int lane, float rfx, float rfy, uint32_t *hdr) {
    constexpr int TH = 4 * R;
    constexpr int STAGE_MAX = R == 8 ? 3 : 2;
    constexpr bool RS = MODE >= MAP_RS_CREATEMAP_CL;          // per-row rotation
    constexpr int BASE = RS ? MODE - MAP_RS_CREATEMAP_CL : MODE;  // the projection pair
    const WarpArgs &a = ta.w;
    const int x0 = tile_x * 64, y0 = tile_y * TH;
    // ---- probe: the map on 64 perimeter pixels -> source bounding box of the tile --------------------------
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
#ifdef VSTAB_DEV
        if (ta.ablate & 1) ax = (float)(x0 + px) * (32.0f * (float)a.sw / (float)a.dw), ay = (float)(y0 + py) * (32.0f * (float)a.sh / (float)a.dh);
#endif
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
    const int lox = max(mnx - 1, -1), hix = min(mxx + 2, a.sw), loy = max(mny - 1, -1), hiy = min(mxy + 2, a.sh);
    const int bx0 = lox & ~7, by0 = loy & ~1;  // -8 / -2 when the box starts left of / above the source
    const int wb = (hix + 1 - bx0 + 7) & ~7, hb = (hiy + 1 - by0 + 1) & ~1;
    const bool have = mnx < a.sw && mxx >= -1 && mny < a.sh && mxy >= -1;  // else every pixel of the tile is outside
    // staged as whole 8 x 2 blocks with aligned 8-byte loads: unaligned planes, and boxes that reach the last,
    // partial block column of a source whose width is not a multiple of 8, are sampled straight from global memory
    const bool use_lds = have && a.sw >= 8 && wb * hb <= ta.lds_capacity_px && (wb >> 3) * (hb >> 1) <= STAGE_MAX * 256 && ta.src_vec_ok &&
                         ((a.sw & 7) == 0 || bx0 + wb <= (a.sw & ~7));
    if (lane == 0) {
        *reinterpret_cast<uint4 *>(hdr) = make_uint4((uint32_t)bx0, (uint32_t)by0, (uint32_t)wb, (uint32_t)hb);
        *reinterpret_cast<uint2 *>(hdr + 4) = make_uint2(use_lds ? 1u : 0u, (uint32_t)(tile_y * ta.tiles_x + tile_x));
    }
}

template <int R, int MODE, int FMT, bool CACHED>
__global__ void __launch_bounds__(256) k_warp_fused(FusedArgs ta) {
    constexpr int TH = 4 * R;                  // tile height
    constexpr int STAGE_MAX = R == 8 ? 3 : 2;  // trips of the staging loop: STAGE_MAX * 256 blocks of 8 x 2 source pixels cover the LDS budget
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *const tile = smem + 8;  // smem[0..5]: the tile header (box, flag, tile index), written by wave 0
    const WarpArgs &a = ta.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float rfx = rcp_refined(a.p.ofx), rfy = rcp_refined(a.p.ofy);
    constexpr int QB = CACHED ? 0 : QMAGIC_BITS;  // offset of the quantised-coordinate representation kept in registers
    constexpr bool RS = MODE >= MAP_RS_CREATEMAP_CL;          // per-row rotation (BASELINE config 5)
    constexpr int BASE = RS ? MODE - MAP_RS_CREATEMAP_CL : MODE;  // the projection pair

    // ---- tile of this workgroup.  Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8), so every XCD
    // gets one contiguous band of tile rows in raster order: horizontally and vertically adjacent tiles then share an
    // L2 and the 128-B lines their source and output rows straddle move once (cutting the image into 16 chunks and
    // pairing a central with an outer chunk per XCD, to balance the larger source boxes of the centre, ran no faster and
    // fetched 20 % more).  Placement only affects speed, never results.
    int tile_x, tile_y;
    {
        const int k = (int)(blockIdx.x & 7u);
        const int idx = (int)(blockIdx.x >> 3);
        const int r0 = (k * ta.tiles_y) >> 3, r1 = ((k + 1) * ta.tiles_y) >> 3;
        if (idx >= (r1 - r0) * ta.tiles_x) return;  // uniform for the workgroup (before any barrier)
        tile_y = r0 + idx / ta.tiles_x;
        tile_x = idx - (tile_y - r0) * ta.tiles_x;
    }
    // ---- probe (wave 0 only; the other waves wait at the barrier without taking issue slots) --------------------
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);  // three waves wait for this one
        probe_tile<R, MODE, CACHED>(ta, tile_x, tile_y, lane, rfx, rfy, smem);
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    const int bx0 = __builtin_amdgcn_readfirstlane((int)smem[0]), by0 = __builtin_amdgcn_readfirstlane((int)smem[1]);
    const int wb = __builtin_amdgcn_readfirstlane((int)smem[2]), hb = __builtin_amdgcn_readfirstlane((int)smem[3]);
    const bool use_lds = __builtin_amdgcn_readfirstlane((int)smem[4]) != 0;
    const int x0 = tile_x * 64, y0 = tile_y * TH;
    const int x = x0 + lane;
#ifdef VSTAB_DEV
    unsigned long long t_rt0 = 0, t_ck0 = 0;
    if (ta.timing) t_rt0 = __builtin_amdgcn_s_memrealtime(), t_ck0 = __builtin_amdgcn_s_memtime();
#endif

    // ---- load: this thread's 8x2 blocks of the box, all loads in flight at once --------------------------------
    const int ux_n = wb >> 3, units = use_lds ? ux_n * (hb >> 1) : 0;
    uint2 y0w[STAGE_MAX], y1w[STAGE_MAX], uvw[STAGE_MAX];
    int ldsoff[STAGE_MAX];  // dword offset of the block in the LDS tile (| ZERO_BLOCK: outside the source); -1 = no block
    constexpr int ZERO_BLOCK = 1 << 24;
    if (use_lds) {
        // (uy, ux) = divmod(tid, ux_n), then +256 blocks per trip; the float quotient is off by at most one
        const float rn = __builtin_amdgcn_rcpf((float)ux_n);
        int uy = (int)((float)tid * rn), ux = tid - uy * ux_n;
        if (ux < 0) ux += ux_n, uy--;
        if (ux >= ux_n) ux -= ux_n, uy++;
        int sy_ = (int)(256.0f * rn), sx_ = 256 - sy_ * ux_n;  // uniform: divmod(256, ux_n)
        if (sx_ < 0) sx_ += ux_n, sy_--;
        if (sx_ >= ux_n) sx_ -= ux_n, sy_++;
        const uint32_t pitch_y = (uint32_t)a.pitch_y, pitch_uv = (uint32_t)a.pitch_uv;  // < 2^24, frame < 4 GiB (host check)
        // no branch around any load, so that all of them are in flight under the map phase
#pragma unroll
        for (int it = 0; it < STAGE_MAX; it++) {
            // a thread without a block in this trip, or with a block outside the source (the zero border), loads from
            // the nearest block inside; neither uses what it loaded
            const bool valid = tid + it * 256 < units;
            const int gx = bx0 + 8 * ux, gy = by0 + 2 * uy;
            const bool inside = (uint32_t)gx < (uint32_t)(a.sw & ~7) && (uint32_t)gy < (uint32_t)a.sh;
            const uint32_t cx = (uint32_t)min(max(gx, 0), (a.sw & ~7) - 8), cy = (uint32_t)min(max(gy, 0), a.sh - 2);
            const uint32_t oy = __umul24(cy, pitch_y) + cx, ouv = __umul24(cy >> 1, pitch_uv) + cx;
            y0w[it] = *reinterpret_cast<const uint2 *>(a.y + oy);
            y1w[it] = *reinterpret_cast<const uint2 *>(a.y + oy + pitch_y);
            uvw[it] = *reinterpret_cast<const uint2 *>(a.uv + ouv);
            ldsoff[it] = valid ? (__mul24(2 * uy, wb) + 8 * ux) | (inside ? 0 : ZERO_BLOCK) : -1;
            ux += sx_, uy += sy_;
            if (ux >= ux_n) ux -= ux_n, uy++;
        }
    } else {
#pragma unroll
        for (int it = 0; it < STAGE_MAX; it++) y0w[it] = y1w[it] = uvw[it] = make_uint2(0, 0), ldsoff[it] = -1;
    }

    // ---- map: R exact evaluations per thread (lane = column, rows y0 + wave * R + j) ---------------------------
    int qxb[R], qyb[R];  // quantised coordinates + QB
    if constexpr (CACHED) {
#pragma unroll
        for (int j = 0; j < R; j++) {
            const int y = min(y0 + wave * R + j, a.dh - 1);
            const int2 q = ta.qmap[(size_t)y * ta.qpitch + min(x, ta.qpitch - 1)];
            qxb[j] = q.x, qyb[j] = q.y;
        }
    } else {
        float vx = norm_coord<BASE>((float)x - a.p.ocx, a.p.ofx, rfx);
        // the compiler would sink each map evaluation to its use behind the barrier; these two statements pin the map
        // phase between the loads above (memory clobber) and the conversion below (the coordinates pass through)
        asm volatile("" : "+v"(vx) : : "memory");
        const ColTerm ct = {a.p.r[0] * vx, a.p.r[3] * vx, a.p.r[6] * vx};
        // row terms: lane l < R evaluates row l of this wave once; every lane then reads them from that lane
        const int y_l = y0 + wave * R + (lane & (R - 1));
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
#ifdef VSTAB_DEV
        if (ta.ablate & 1) {
#pragma unroll
            for (int j = 0; j < R; j++) {
                const float fx = (float)x * (32.0f * (float)a.sw / (float)a.dw), fy = (float)(y0 + wave * R + j) * (32.0f * (float)a.sh / (float)a.dh);
                qxb[j] = __float_as_int(fx + QMAGIC), qyb[j] = __float_as_int(fy + QMAGIC);
            }
        } else
#endif
        if constexpr (BASE == MAP_CREATEMAP_CL || BASE == MAP_FISH_TO_RECT) {
            // two rows at a time through the packed-fp32 pipe
#pragma unroll
            for (int j = 0; j < R; j += 2) {
                const f32x2 b0 = {bcast(b0_l, j), bcast(b0_l, j + 1)}, b1 = {bcast(b1_l, j), bcast(b1_l, j + 1)}, b2 = {bcast(b2_l, j), bcast(b2_l, j + 1)};
                f32x2 a0 = splat2(ct.a0), a1 = splat2(ct.a1), a2 = splat2(ct.a2), r02 = splat2(a.p.r[2]), r12 = splat2(a.p.r[5]), r22 = splat2(a.p.r[8]);
                if constexpr (RS) {  // every row has its own matrix: column products and third column per row
                    a0 = (f32x2){bcast(m_l[0], j), bcast(m_l[0], j + 1)} * splat2(vx);
                    a1 = (f32x2){bcast(m_l[3], j), bcast(m_l[3], j + 1)} * splat2(vx);
                    a2 = (f32x2){bcast(m_l[6], j), bcast(m_l[6], j + 1)} * splat2(vx);
                    r02 = (f32x2){bcast(m_l[2], j), bcast(m_l[2], j + 1)};
                    r12 = (f32x2){bcast(m_l[5], j), bcast(m_l[5], j + 1)};
                    r22 = (f32x2){bcast(m_l[8], j), bcast(m_l[8], j + 1)};
                }
                f32x2 ax, ay;
                map_pixel32_x2<BASE == MAP_FISH_TO_RECT>(icx32, icy32, ifx32, ify32, r02, r12, r22, a0, a1, a2, b0, b1, b2, ax, ay);
                ax += splat2(QMAGIC), ay += splat2(QMAGIC);
                qxb[j] = __float_as_int(ax.x), qxb[j + 1] = __float_as_int(ax.y);
                qyb[j] = __float_as_int(ay.x), qyb[j + 1] = __float_as_int(ay.y);
            }
        } else {
#pragma unroll
            for (int j = 0; j < R; j++) {
                const float vy = bcast(vy_l, j);
                const RowTerm rt = {bcast(b0_l, j), bcast(b1_l, j), bcast(b2_l, j)};
                float ax, ay;
                map_pixel_ex<BASE>(ta.p32, a.p, ct, rt, vx, vy, ax, ay);
                qxb[j] = __float_as_int(ax + QMAGIC), qyb[j] = __float_as_int(ay + QMAGIC);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < R; j++) asm volatile("" : "+v"(qxb[j]), "+v"(qyb[j]) : : "memory");

    // ---- convert: cvtColor once per source pixel, BGRx dwords to LDS --------------------------------------------
    if (use_lds) {
#pragma unroll
        for (int it = 0; it < STAGE_MAX; it++) {
            if (ldsoff[it] >= ZERO_BLOCK) {
                uint32_t *d = tile + (ldsoff[it] - ZERO_BLOCK);
                const uint4 z = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4 *>(d) = z, *reinterpret_cast<uint4 *>(d + 4) = z;
                *reinterpret_cast<uint4 *>(d + wb) = z, *reinterpret_cast<uint4 *>(d + wb + 4) = z;
#ifdef VSTAB_DEV
            } else if (ldsoff[it] >= 0 && (ta.ablate & 4)) {
                uint32_t *d = tile + ldsoff[it];
                *reinterpret_cast<uint4 *>(d) = make_uint4(y0w[it].x, y0w[it].y, uvw[it].x, uvw[it].y);
                *reinterpret_cast<uint4 *>(d + 4) = make_uint4(y0w[it].y, y0w[it].x, uvw[it].x, uvw[it].y);
                *reinterpret_cast<uint4 *>(d + wb) = make_uint4(y1w[it].x, y1w[it].y, uvw[it].x, uvw[it].y);
                *reinterpret_cast<uint4 *>(d + wb + 4) = make_uint4(y1w[it].y, y1w[it].x, uvw[it].y, uvw[it].x);
#endif
            } else if (ldsoff[it] >= 0) {
                uint32_t *d = tile + ldsoff[it];
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const uint32_t cw = half ? uvw[it].y : uvw[it].x, ya = half ? y0w[it].y : y0w[it].x, yb = half ? y1w[it].y : y1w[it].x;
                    const ChromaTerm c0 = chroma_term_folded(cw & 255, (cw >> 8) & 255);
                    const ChromaTerm c1 = chroma_term_folded((cw >> 16) & 255, cw >> 24);
                    uint4 r0, r1;
                    r0.x = pack_bgrx(ya & 255, c0), r0.y = pack_bgrx((ya >> 8) & 255, c0);
                    r0.z = pack_bgrx((ya >> 16) & 255, c1), r0.w = pack_bgrx(ya >> 24, c1);
                    r1.x = pack_bgrx(yb & 255, c0), r1.y = pack_bgrx((yb >> 8) & 255, c0);
                    r1.z = pack_bgrx((yb >> 16) & 255, c1), r1.w = pack_bgrx(yb >> 24, c1);
                    *reinterpret_cast<uint4 *>(d + 4 * half) = r0;
                    *reinterpret_cast<uint4 *>(d + wb + 4 * half) = r1;
                }
            }
        }
    }
    __syncthreads();

    // ---- sample + blend -----------------------------------------------------------------------------------------
    const bool col_live = x < a.dw;
    uint32_t out[R];
    uint32_t slow = 0;  // bit j: live pixel whose footprint is not inside the staged box
    {
        const int cx = bx0 + (QB >> 5), cy = by0 + (QB >> 5);
        const uint32_t wlim = use_lds ? (uint32_t)(wb - 1) : 0u, hlim = (uint32_t)(hb - 1);  // both taps of each axis inside the staged box
#pragma unroll
        for (int j = 0; j < R; j++) {
            const int Xr = (qxb[j] >> 5) - cx, Yr = (qyb[j] >> 5) - cy;
            const bool inbox = (uint32_t)Xr < wlim && (uint32_t)Yr < hlim;
            uint32_t v = 0;
            if (inbox) {
                const uint32_t *t = tile + (__mul24(Yr, wb) + Xr);
#ifdef VSTAB_DEV
                if (ta.ablate & 2) v = t[0] ^ t[1] ^ t[wb] ^ t[wb + 1] ^ (qxb[j] & 31) ^ (qyb[j] & 31);
                else
#endif
                v = blend_bgrx(t[0], t[1], t[wb], t[wb + 1], qxb[j] & 31, qyb[j] & 31);
            }
            out[j] = v;
            const bool live = col_live && y0 + wave * R + j < a.dh;
            slow |= (!inbox && live ? 1u : 0u) << j;
        }
    }
    if (__builtin_amdgcn_ballot_w64(slow != 0)) {  // rare: source border, degenerate rotation, box over the LDS budget
#pragma unroll 1
        for (int j = 0; j < R; j++) {
            int sx = qxb[0], sy = qyb[0];
#pragma unroll
            for (int k = 1; k < R; k++) sx = j == k ? qxb[k] : sx, sy = j == k ? qyb[k] : sy;
            if ((slow >> j) & 1u) {
                const uint32_t v = gather_pixel_far(a, sx - QB, sy - QB);
#pragma unroll
                for (int k = 0; k < R; k++) out[k] = j == k ? v : out[k];
            }
        }
    }

    // ---- store: a row of 64 BGRx dwords -> 48 dwords of BGR, one per lane (ds_bpermute transposition) ----------
#ifdef VSTAB_DEV
    if (ta.ablate & 8) {
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < R; j++) acc ^= out[j];
        if (acc == 0x12345678u) a.dst[0] = 1;
        return;
    }
#endif
    const int ncols = min(64, a.dw - x0);  // > 0
    if constexpr (FMT == 0) {
        const int p0 = (4 * lane) / 3, m3 = lane - 3 * (lane / 3);  // pixels p0, p0 + 1 feed dword `lane` (lane < 48)
        const uint32_t sel = m3 == 0 ? 0x04020100u : m3 == 1 ? 0x05040201u : 0x06050402u;
        const int nbytes = 3 * ncols, nfull = nbytes >> 2, rem = nbytes & 3;
#pragma unroll
        for (int j = 0; j < R; j++) {
            const int y = y0 + wave * R + j;
            if (y >= a.dh) break;  // uniform
            uint8_t *o = a.dst + ((size_t)(uint32_t)y * a.pitch_dst + (uint32_t)x0 * 3u);  // uniform
            if (ta.dst_vec_ok) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * p0, (int)out[j]);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * p0 + 4, (int)out[j]);
                const uint32_t d = __builtin_amdgcn_perm(hi, lo, sel);
                if (lane < nfull) *reinterpret_cast<uint32_t *>(o + (uint32_t)(4 * lane)) = d;
                else if (lane == nfull && rem) {
                    for (int i = 0; i < rem; i++) o[4 * lane + i] = (d >> (8 * i)) & 255;
                }
            } else if (col_live) {
                o[3 * lane] = out[j] & 255, o[3 * lane + 1] = (out[j] >> 8) & 255, o[3 * lane + 2] = (out[j] >> 16) & 255;
            }
        }
    } else {
        // NV12 output: luma for every pixel, chroma from the even-row / even-column pixels; four lanes' bytes are
        // collected in the first lane of each quad (DPP quad_perm) and stored as one dword
#pragma unroll
        for (int j = 0; j < R; j++) {
            const int y = y0 + wave * R + j;
            if (y >= a.dh) break;  // uniform
            const uint32_t yb = bgr_to_y(out[j]);
            const uint32_t y1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)yb, 0x55, 0xf, 0xf, true);
            const uint32_t y2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)yb, 0xaa, 0xf, 0xf, true);
            const uint32_t y3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)yb, 0xff, 0xf, 0xf, true);
            const uint32_t yw = yb | (y1 << 8) | (y2 << 16) | (y3 << 24);
            uint8_t *o = a.dst + ((size_t)(uint32_t)y * a.pitch_dst + (uint32_t)x);
            if (!(lane & 3) && col_live) {
                if (ta.dst_vec_ok && x + 4 <= a.dw) {
                    *reinterpret_cast<uint32_t *>(o) = yw;
                } else {
                    for (int i = 0; i < 4 && x + i < a.dw; i++) o[i] = (yw >> (8 * i)) & 255;
                }
            }
            if (!(y & 1)) {
                const uint32_t cb = bgr_to_uv(out[j]);
                const uint32_t c2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)cb, 0xaa, 0xf, 0xf, true);
                const uint32_t cw = cb | (c2 << 16);
                uint8_t *c = a.dst_uv + ((size_t)(uint32_t)(y >> 1) * a.pitch_dst_uv + (uint32_t)x);
                if (!(lane & 3) && col_live) {
                    if (ta.dst_vec_ok && x + 2 < a.dw) {
                        *reinterpret_cast<uint32_t *>(c) = cw;
                    } else {
                        for (int i = 0; i < 4 && x + (i & ~1) < a.dw; i++) c[i] = (cw >> (8 * i)) & 255;
                    }
                }
            }
        }
    }
#ifdef VSTAB_DEV
    if (ta.timing && tid == 0) {
        unsigned long long *t = ta.timing + 4 * (size_t)(tile_y * ta.tiles_x + tile_x);
        t[0] = t_rt0, t[1] = t_ck0, t[2] = __builtin_amdgcn_s_memrealtime(), t[3] = __builtin_amdgcn_s_memtime();
    }
#endif
}

}  // namespace vstab

using namespace vstab;

// Launch of the fused kernel; called by warp_impl (vstab_warp.hip) after argument validation.
namespace vstab {
#ifdef VSTAB_DEV
static unsigned long long *g_dev_timing = nullptr;
extern "C" __attribute__((visibility("default"))) void vstab_dev_set_timing(void *p) { g_dev_timing = static_cast<unsigned long long *>(p); }
#endif

vstab_status launch_warp_fused(const WarpArgs &a, const float params[17], int map_mode, bool nv12_out, bool src_vec_ok, bool dst_vec_ok,
                               const void *qmap, int qpitch, const float *rot_bottom, hipStream_t st) {
    FusedArgs ta;
    ta.w = a;
    ta.p32 = {params[0] * 32.0f, params[1] * 32.0f, params[2] * 32.0f, params[3] * 32.0f, params[10], params[13], params[16]};
    ta.src_vec_ok = src_vec_ok, ta.dst_vec_ok = dst_vec_ok;
    ta.qmap = static_cast<const int2 *>(qmap), ta.qpitch = qpitch;
    for (int k = 0; k < 9; k++) ta.rs_d[k] = rot_bottom ? rot_bottom[k] - params[8 + k] : 0.0f;  // fp32, as the oracle forms it
    ta.rs_den = (float)(a.dh > 1 ? a.dh - 1 : 1);
    if (rot_bottom) map_mode += MAP_RS_CREATEMAP_CL;  // modes 0 / 1 only (checked by the caller)
#ifdef VSTAB_DEV
    ta.timing = g_dev_timing;
    static const int ablate = getenv("VSTAB_ABLATE") ? atoi(getenv("VSTAB_ABLATE")) : 0;
    ta.ablate = ablate;
#endif
    // Tile shape: 64 x 32 output pixels and 40 KB of LDS (4 workgroups per CU) when that gives the 1024 workgroup
    // slots of the chip a few rounds of tiles; 64 x 16 with 24 KB for small outputs (1080p).
    const long tiles32 = (long)div_up(a.dw, 64) * div_up(a.dh, 32);
    int rows = tiles32 < 1536 ? 4 : 8, lds_kb = rows == 4 ? 24 : 40;
#ifdef VSTAB_DEV
    if (const char *e = getenv("VSTAB_ROWS")) rows = atoi(e) == 4 ? 4 : 8;
    if (const char *e = getenv("VSTAB_LDS_KB")) lds_kb = atoi(e);
#endif
    const size_t lds_bytes = (size_t)lds_kb * 1024;
    ta.lds_capacity_px = (int)(lds_bytes / 4) - 8;  // 8 dwords hold the tile header
    ta.tiles_x = (int)div_up(a.dw, 64), ta.tiles_y = (int)div_up(a.dh, 4 * rows);
    // every XCD owns one band of tile rows (see the kernel); the grid holds the largest of the eight shares per XCD
    int share = 0;
    for (int k = 0; k < 8; k++) share = std::max(share, ((((k + 1) * ta.tiles_y) >> 3) - ((k * ta.tiles_y) >> 3)) * ta.tiles_x);
    const dim3 grid(8u * (unsigned)share);
#define VSTAB_LAUNCH(R, M, F, C) hipLaunchKernelGGL((k_warp_fused<R, M, F, C>), grid, dim3(256), lds_bytes, st, ta)
#define VSTAB_LAUNCH_RF(M, C)                                  \
    do {                                                       \
        if (rows == 8 && !nv12_out) VSTAB_LAUNCH(8, M, 0, C);  \
        else if (rows == 8) VSTAB_LAUNCH(8, M, 1, C);          \
        else if (!nv12_out) VSTAB_LAUNCH(4, M, 0, C);          \
        else VSTAB_LAUNCH(4, M, 1, C);                         \
    } while (0)
    if (qmap) {  // the map phase reads the quantised map: the map mode no longer matters
        VSTAB_LAUNCH_RF(MAP_CREATEMAP_CL, true);
    } else {
        switch (map_mode) {
            case VSTAB_MAP_CREATEMAP_CL: VSTAB_LAUNCH_RF(MAP_CREATEMAP_CL, false); break;
            case VSTAB_MAP_FISH_TO_RECT: VSTAB_LAUNCH_RF(MAP_FISH_TO_RECT, false); break;
            case VSTAB_MAP_FISH_TO_FISH: VSTAB_LAUNCH_RF(MAP_FISH_TO_FISH, false); break;
            case VSTAB_MAP_RECT_TO_RECT: VSTAB_LAUNCH_RF(MAP_RECT_TO_RECT, false); break;
            case VSTAB_MAP_RECT_TO_FISH: VSTAB_LAUNCH_RF(MAP_RECT_TO_FISH, false); break;
            case VSTAB_MAP_CREATEMAP_CL_OPENCL: VSTAB_LAUNCH_RF(MAP_CREATEMAP_CL_OPENCL, false); break;
            case MAP_RS_CREATEMAP_CL: VSTAB_LAUNCH_RF(MAP_RS_CREATEMAP_CL, false); break;
            default: VSTAB_LAUNCH_RF(MAP_RS_FISH_TO_RECT, false); break;
        }
    }
#undef VSTAB_LAUNCH_RF
#undef VSTAB_LAUNCH
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}

}  // namespace vstab
