// vstab_warp_fused.hip -- the fused undistort-remap kernel of the hot path (gfx950 / MI355X).
//
//   createMap.cl:13-50  ->  cv::remap quantisation (FrameSourceWarp.cpp:306-312)  ->  NV12 taps converted with the
//   cvtColor arithmetic (:401)  ->  fixed-point bilinear blend  ->  BGR8 (or NV12).
// Converting each tap and then blending is exactly what the reference's cvtColor-then-remap sequence computes, so the
// output is bit-identical to the un-fused operators.  Compiled with -ffp-contract=off.
//
// One workgroup (4 waves) per output tile of 64 columns x 4 RW rows; lane = output column, a wave owns RW consecutive
// rows.  A launch mixes two tile heights: RW = RWB for the bulk of the image and RW = RWB / 2 for the last rows of every
// XCD's band, so that the last round of workgroups -- which cannot fill the chip -- is made of tiles half as long.
//   probe    wave 0 evaluates the map on 64 perimeter pixels of the tile (approximate arithmetic) and reduces them (DPP)
//            to the tile's source bounding box.  A continuous map attains its coordinate extremes on the perimeter.
//   load     all global loads of the thread's share of the box (8 x 2 blocks: two luma rows + one chroma row, 8-byte
//            loads) are issued at once ...
//   map      ... and the RW exact map evaluations of the thread run underneath their latency: two rows per packed-fp32
//            instruction, the RW / 2 row pairs advanced in lock-step so that no instruction waits for its predecessor.
//   convert  the loaded blocks are converted ONCE per source pixel to BGRx with the cvtColor fixed-point arithmetic and
//            written to LDS (ds_write_b128); blocks outside the source become the zero border of BORDER_CONSTANT.
//   sample   when every footprint of the wave lies inside the staged box (the normal case, one wave-uniform test) all
//            2 RW tap-pair reads of a thread are issued before the first blend; otherwise pixel by pixel with the
//            global-memory gather for footprints outside the box.
//   store    rows transposed to 4-byte-per-lane stores with ds_bpermute.
// Bit-exactness never depends on the box: a pixel whose 2x2 footprint is not inside the staged box (degenerate
// rotations, a box larger than the LDS budget, unaligned planes) is sampled straight from global memory with per-tap
// zeroing (gather_pixel), which computes the same integers.  HBM traffic = the NV12 frame once + the output once.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>

#include <hip/hip_ext.h>

#include "vstab_device.hpp"
#include "vstab_device10.hpp"
#include "vstab_internal.hpp"
#include "vstab_warp_args.hpp"
#include "vstab_warp_tile.hpp"

namespace vstab {

// ---------------------------------------------------------------------------------------------------------------------
// One output tile: 64 columns x 4 RW rows at (x0, y0).  All 256 threads of the workgroup take part.
// ---------------------------------------------------------------------------------------------------------------------
#ifdef VSTAB_DEV
// development builds: eight 100 MHz wall-clock stamps per wave (tools/wg_timeline.py)
#define VSTAB_STAMP(k)                                                                                                   \
    if (ta.timing && lane == 0) ta.timing[((size_t)blockIdx.x * 4 + (size_t)wave) * 8 + (k)] = __builtin_amdgcn_s_memrealtime()
#else
#define VSTAB_STAMP(k)
#endif

// Returns false -- right behind the probe, before anything else is done -- when SPLIT is set and the tile's box is over the
// LDS budget: the caller then covers the tile with two tiles of half the height.
// DEPTH 8: NV12 bytes in, the reference's pixel path.  DEPTH 10 (FMT 2, BASELINE config 5): P010 words in, the box staged as
// B | G << 10 | R << 20 dwords -- the same 4-byte LDS pixel -- blended by BLEND (vstab_device10.hpp), 16-bit BGR out.
template <int RWB, int RW, int MODE, int FMT, bool CACHED, bool SPLIT, int DEPTH = 8, int BLEND = 0>
__device__ __forceinline__ bool warp_tile(const FusedArgs &ta, uint32_t *smem, const int x0, const int y0) {
    static_assert((DEPTH == 8 && FMT < 2) || (DEPTH == 10 && FMT >= 2), "10-bit pixels leave as 16-bit BGR (FMT 2) or P010 planes (FMT 3)");
    using SrcVec = typename std::conditional<DEPTH == 10, uint4, uint2>::type;  // 8 source samples of a block row
    constexpr uint32_t BPS = DEPTH == 10 ? 2 : 1;                              // bytes per sample
    constexpr int TH = 4 * RW;
    constexpr int STAGE_MAX = StageTrips<RWB, RW>::value;
    constexpr int QB = CACHED ? 0 : QMAGIC_BITS;  // offset of the quantised-coordinate representation kept in registers
    // LDS pixel: one dword (BGRx bytes, or B | G << 10 | R << 20), or -- 10-bit pixels with the fp16 blend -- two: the three channels
    // as binary16 numbers, so that the blend's taps are converted once per source pixel (vstab_device10.hpp)
    constexpr bool PIX8 = DEPTH == 10 && BLEND == VSTAB_BLEND_FP16;
    constexpr int PD = PIX8 ? 2 : 1;
    constexpr bool RS = map_mode_is_rs(MODE);    // per-row rotation (BASELINE config 5)
    constexpr int BASE = map_mode_base(MODE);    // the projection pair and its arithmetic
    uint32_t *const tile = smem + 8;  // smem[0..4]: the tile header (box, flag), written by wave 0
    const WarpArgs &a = ta.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float rfx = rcp_refined(a.p.ofx), rfy = rcp_refined(a.p.ofy);

    VSTAB_STAMP(0);
    // ---- probe (wave 0 only; the other waves wait at the barrier without taking issue slots) --------------------
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);  // three waves wait for this one
        probe_tile<TH, STAGE_MAX, MODE, CACHED>(ta, x0, y0, lane, rfx, rfy, smem);
        __builtin_amdgcn_s_setprio(VSTAB_WARP_PRIO);
    }
    __syncthreads();
    const int bx0 = __builtin_amdgcn_readfirstlane((int)smem[0]), by0 = __builtin_amdgcn_readfirstlane((int)smem[1]);
    const int wb = __builtin_amdgcn_readfirstlane((int)smem[2]), hb = __builtin_amdgcn_readfirstlane((int)smem[3]);
    const int box_state = __builtin_amdgcn_readfirstlane((int)smem[4]);
    if constexpr (SPLIT) {
        if (box_state == 2) return false;  // uniform
    }
    const bool use_lds = box_state == 1;
    const int x = x0 + lane;
    VSTAB_STAMP(1);

    // ---- load: this thread's 8x2 blocks of the box, all loads in flight at once --------------------------------
    const int ux_n = wb >> 3, units = use_lds ? ux_n * (hb >> 1) : 0;
#ifdef VSTAB_DEV
    const int pw = (wb + ta.lds_pad) * PD;  // LDS row pitch in dwords (development experiment: pad against bank conflicts)
#else
    const int pw = wb * PD;                 // LDS row pitch in dwords
#endif
    SrcVec y0w[STAGE_MAX], y1w[STAGE_MAX], uvw[STAGE_MAX];
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
        // no per-lane branch around any load, so that all of them are in flight under the map phase; a trip that no
        // thread of the workgroup needs (most boxes have fewer than 256 blocks) is skipped by a scalar branch: its
        // ~30 vector instructions of index arithmetic per thread were 4 % of the kernel each (35.4 -> 34.2 us alone at 4K)
#pragma unroll
        for (int it = 0; it < STAGE_MAX; it++) {
            if (it > 0 && units <= it * 256) {  // uniform
                y0w[it] = y1w[it] = uvw[it] = SrcVec(), ldsoff[it] = -1;
                continue;
            }
            // a thread without a block in this trip, or with a block outside the source (the zero border), loads from
            // the nearest block inside; neither uses what it loaded
            const bool valid = tid + it * 256 < units;
            const int gx = bx0 + 8 * ux, gy = by0 + 2 * uy;
            const bool inside = (uint32_t)gx < (uint32_t)(a.sw & ~7) && (uint32_t)gy < (uint32_t)a.sh;
            const uint32_t cx = (uint32_t)min(max(gx, 0), (a.sw & ~7) - 8), cy = (uint32_t)min(max(gy, 0), a.sh - 2);
            const uint32_t oy = __umul24(cy, pitch_y) + cx * BPS, ouv = __umul24(cy >> 1, pitch_uv) + cx * BPS;
            y0w[it] = *reinterpret_cast<const SrcVec *>(a.y + oy);
            y1w[it] = *reinterpret_cast<const SrcVec *>(a.y + (oy + pitch_y));  // (a 32-bit offset: one add, not a 64-bit pointer step)
            uvw[it] = *reinterpret_cast<const SrcVec *>(a.uv + ouv);
            ldsoff[it] = valid ? (__mul24(2 * uy, pw) + 8 * PD * ux) | (inside ? 0 : ZERO_BLOCK) : -1;
            ux += sx_, uy += sy_;
            if (ux >= ux_n) ux -= ux_n, uy++;
        }
    } else {
#pragma unroll
        for (int it = 0; it < STAGE_MAX; it++) y0w[it] = y1w[it] = uvw[it] = SrcVec(), ldsoff[it] = -1;
    }

    VSTAB_STAMP(2);
    // ---- map: RW exact evaluations per thread (lane = column, rows y0 + wave * RW + j): vstab_warp_tile.hpp -----
    int qxb[RW], qyb[RW];  // quantised coordinates + QB
    {
        int unused_cx[RW / 2], unused_cy[RW / 2];
        const float unused_qm[4] = {QMAGIC, QMAGIC, QMAGIC, QMAGIC};
        map_phase<RW, MODE, CACHED, false>(ta, x, y0, wave, lane, rfx, rfy, qxb, qyb, unused_cx, unused_cy, unused_qm);
    }
#pragma unroll
    for (int j = 0; j < RW; j++) asm volatile("" : "+v"(qxb[j]), "+v"(qyb[j]) : : "memory");

    VSTAB_STAMP(3);
    // ---- convert: cvtColor once per source pixel, BGRx dwords to LDS --------------------------------------------
    if (use_lds) {
#pragma unroll
        for (int it = 0; it < STAGE_MAX; it++) {
            if (ldsoff[it] >= ZERO_BLOCK) {
                uint32_t *d = tile + (ldsoff[it] - ZERO_BLOCK);
                const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2 * PD; q++) *reinterpret_cast<uint4 *>(d + 4 * q) = z, *reinterpret_cast<uint4 *>(d + pw + 4 * q) = z;
#ifdef VSTAB_DEV
            } else if (DEPTH == 8 && ldsoff[it] >= 0 && (ta.ablate & 4)) {
                uint32_t *d = tile + ldsoff[it];
                *reinterpret_cast<uint4 *>(d) = make_uint4(y0w[it].x, y0w[it].y, uvw[it].x, uvw[it].y);
                *reinterpret_cast<uint4 *>(d + 4) = make_uint4(y0w[it].y, y0w[it].x, uvw[it].x, uvw[it].y);
                *reinterpret_cast<uint4 *>(d + pw) = make_uint4(y1w[it].x, y1w[it].y, uvw[it].x, uvw[it].y);
                *reinterpret_cast<uint4 *>(d + pw + 4) = make_uint4(y1w[it].y, y1w[it].x, uvw[it].y, uvw[it].x);
#endif
            } else if (ldsoff[it] >= 0) {
                uint32_t *d = tile + ldsoff[it];
                if constexpr (DEPTH == 10) {
                    // eight P010 luma words per row (two per dword), four (U, V) word pairs; sample = word >> 6
                    const uint32_t ya[4] = {y0w[it].x, y0w[it].y, y0w[it].z, y0w[it].w}, yb[4] = {y1w[it].x, y1w[it].y, y1w[it].z, y1w[it].w};
                    const uint32_t cw[4] = {uvw[it].x, uvw[it].y, uvw[it].z, uvw[it].w};
#pragma unroll
                    for (int half = 0; half < 2; half++) {
                        const ChromaTerm c0 = chroma_term10(cw[2 * half]), c1 = chroma_term10(cw[2 * half + 1]);
                        if constexpr (PIX8) {  // four pixels of each row as halves: two 16-byte stores per row
                            const uint32_t la = ya[2 * half], lb = ya[2 * half + 1], ma = yb[2 * half], mb = yb[2 * half + 1];
                            const uint2 a0 = pack_bgr10h((int)((la & 0xffffu) >> 6), c0), a1 = pack_bgr10h((int)(la >> 22), c0);
                            const uint2 a2 = pack_bgr10h((int)((lb & 0xffffu) >> 6), c1), a3 = pack_bgr10h((int)(lb >> 22), c1);
                            const uint2 b0 = pack_bgr10h((int)((ma & 0xffffu) >> 6), c0), b1 = pack_bgr10h((int)(ma >> 22), c0);
                            const uint2 b2 = pack_bgr10h((int)((mb & 0xffffu) >> 6), c1), b3 = pack_bgr10h((int)(mb >> 22), c1);
                            *reinterpret_cast<uint4 *>(d + 8 * half) = make_uint4(a0.x, a0.y, a1.x, a1.y);
                            *reinterpret_cast<uint4 *>(d + 8 * half + 4) = make_uint4(a2.x, a2.y, a3.x, a3.y);
                            *reinterpret_cast<uint4 *>(d + pw + 8 * half) = make_uint4(b0.x, b0.y, b1.x, b1.y);
                            *reinterpret_cast<uint4 *>(d + pw + 8 * half + 4) = make_uint4(b2.x, b2.y, b3.x, b3.y);
                            continue;
                        }
                        uint4 r0, r1;
                        r0.x = pack_bgr10((int)((ya[2 * half] & 0xffffu) >> 6), c0), r0.y = pack_bgr10((int)(ya[2 * half] >> 22), c0);
                        r0.z = pack_bgr10((int)((ya[2 * half + 1] & 0xffffu) >> 6), c1), r0.w = pack_bgr10((int)(ya[2 * half + 1] >> 22), c1);
                        r1.x = pack_bgr10((int)((yb[2 * half] & 0xffffu) >> 6), c0), r1.y = pack_bgr10((int)(yb[2 * half] >> 22), c0);
                        r1.z = pack_bgr10((int)((yb[2 * half + 1] & 0xffffu) >> 6), c1), r1.w = pack_bgr10((int)(yb[2 * half + 1] >> 22), c1);
                        *reinterpret_cast<uint4 *>(d + 4 * half) = r0;
                        *reinterpret_cast<uint4 *>(d + pw + 4 * half) = r1;
                    }
                } else
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
                    *reinterpret_cast<uint4 *>(d + pw + 4 * half) = r1;
                }
            }
        }
    }
    VSTAB_STAMP(4);
    __syncthreads();
    VSTAB_STAMP(5);

    // ---- sample + blend -----------------------------------------------------------------------------------------
    const bool col_live = x < a.dw;
    uint32_t out[RW];
    {
        // box origin in the registers' representation, each ONE scalar (left to itself the compiler subtracts the origin and the
        // representation's offset from every coordinate separately: 16 vector instructions per thread)
        int cx = bx0 + (QB >> 5), cy = by0 + (QB >> 5);
        asm("" : "+s"(cx), "+s"(cy));
        const uint32_t wlim = use_lds ? (uint32_t)(wb - 1) : 0u, hlim = (uint32_t)(hb - 1);  // both taps of each axis inside the staged box
        int Xr[RW], Yr[RW];
        const uint32_t wb4 = (uint32_t)pw << 2;
        // LDS byte address of the tile, as a scalar: tap addresses are formed from it by hand (shift-add, multiply-add, add)
        uint32_t tile_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)tile;
        asm("" : "+s"(tile_lds));
        uint32_t mxx = 0, mxy = 0;  // as unsigned: a coordinate left of / above the box is huge
#pragma unroll
        for (int j = 0; j < RW; j++) {
            Xr[j] = (qxb[j] >> 5) - cx, Yr[j] = (qyb[j] >> 5) - cy;
            mxx = max(mxx, (uint32_t)Xr[j]), mxy = max(mxy, (uint32_t)Yr[j]);
        }
        if (!__builtin_amdgcn_ballot_w64(mxx >= wlim || mxy >= hlim)) {
            // every footprint of the wave lies in the staged box: all tap reads first, then the blends
            constexpr int TG = RW < TAP_GROUP ? RW : TAP_GROUP;  // rows whose tap reads are in flight together
            if constexpr (PIX8) {
#pragma unroll
                for (int j0 = 0; j0 < RW; j0 += TG) {
                    uint2 t0[TG], t1[TG], t2[TG], t3[TG];
#pragma unroll
                    for (int j = 0; j < TG; j++) {
                        uint32_t ax8, au;  // (as below: shift-add, multiply-add, add)
                        asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(ax8) : "v"(Xr[j0 + j]), "s"(tile_lds));
                        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(au) : "v"(Yr[j0 + j]), "s"(wb4), "v"(ax8));
                        const LdsPair *u = reinterpret_cast<const LdsPair *>(au), *l = reinterpret_cast<const LdsPair *>(au + wb4);
                        const u32x2 v0 = u[0], v1 = u[1], v2 = l[0], v3 = l[1];
                        t0[j] = make_uint2(v0.x, v0.y), t1[j] = make_uint2(v1.x, v1.y), t2[j] = make_uint2(v2.x, v2.y), t3[j] = make_uint2(v3.x, v3.y);
                    }
#pragma unroll
                    for (int j = 0; j < TG; j++) out[j0 + j] = blend_bgr10h(t0[j], t1[j], t2[j], t3[j], qxb[j0 + j] & 31, qyb[j0 + j] & 31);
                }
            } else
#pragma unroll
            for (int j0 = 0; j0 < RW; j0 += TG) {
                uint32_t t0[TG], t1[TG], t2[TG], t3[TG];
#pragma unroll
                for (int j = 0; j < TG; j++) {
                    // LDS addresses by hand: shift-add and multiply-add for the upper tap row, one add for the lower (written as
                    // instructions: the compiler re-associates the expression into four)
                    uint32_t ax4, au;
                    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(ax4) : "v"(Xr[j0 + j]), "s"(tile_lds));
                    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(au) : "v"(Yr[j0 + j]), "s"(wb4), "v"(ax4));
                    const LdsWord *u = reinterpret_cast<const LdsWord *>(au), *l = reinterpret_cast<const LdsWord *>(au + wb4);
                    t0[j] = u[0], t1[j] = u[1], t2[j] = l[0], t3[j] = l[1];
                }
#pragma unroll
                for (int j = 0; j < TG; j++) {
#ifdef VSTAB_DEV
                    if (ta.ablate & 2) out[j0 + j] = t0[j] ^ t1[j] ^ t2[j] ^ t3[j] ^ (qxb[j0 + j] & 31) ^ (qyb[j0 + j] & 31);
                    else
#endif
                    out[j0 + j] = DEPTH == 10 ? blend_bgr10<BLEND>(t0[j], t1[j], t2[j], t3[j], qxb[j0 + j] & 31, qyb[j0 + j] & 31)
                                              : blend_bgrx(t0[j], t1[j], t2[j], t3[j], qxb[j0 + j] & 31, qyb[j0 + j] & 31);
                }
            }
        } else {
            // rare (source border with a box that had to be cut, degenerate rotation, box over the LDS budget): pixel by
            // pixel, from the box where the footprint is inside it and from global memory where it is not
            uint32_t slow = 0;  // bit j: live pixel whose footprint is not inside the staged box
#pragma unroll
            for (int j = 0; j < RW; j++) {
                const bool inbox = (uint32_t)Xr[j] < wlim && (uint32_t)Yr[j] < hlim;
                uint32_t v = 0;
                if (inbox) {
                    const uint32_t *t = tile + (__mul24(Yr[j], pw) + PD * Xr[j]);
                    if constexpr (PIX8)
                        v = blend_bgr10h(make_uint2(t[0], t[1]), make_uint2(t[2], t[3]), make_uint2(t[pw], t[pw + 1]), make_uint2(t[pw + 2], t[pw + 3]), qxb[j] & 31,
                                         qyb[j] & 31);
                    else
                        v = DEPTH == 10 ? blend_bgr10<BLEND>(t[0], t[1], t[pw], t[pw + 1], qxb[j] & 31, qyb[j] & 31)
                                        : blend_bgrx(t[0], t[1], t[pw], t[pw + 1], qxb[j] & 31, qyb[j] & 31);
                }
                out[j] = v;
                const bool live = col_live && y0 + wave * RW + j < a.dh;
                slow |= (!inbox && live ? 1u : 0u) << j;
            }
            if (__builtin_amdgcn_ballot_w64(slow != 0)) {
#pragma unroll 1
                for (int j = 0; j < RW; j++) {
                    int sx = qxb[0], sy = qyb[0];
#pragma unroll
                    for (int k = 1; k < RW; k++) sx = j == k ? qxb[k] : sx, sy = j == k ? qyb[k] : sy;
                    if ((slow >> j) & 1u) {
                        const uint32_t v = DEPTH == 10 ? gather_pixel10<BLEND>(a, sx - QB, sy - QB) : gather_pixel_far(a, sx - QB, sy - QB);
#pragma unroll
                        for (int k = 0; k < RW; k++) out[k] = j == k ? v : out[k];
                    }
                }
            }
        }
    }

    VSTAB_STAMP(6);
    // ---- store: a row of 64 BGRx dwords -> 48 dwords of BGR, one per lane (ds_bpermute transposition) ----------
#ifdef VSTAB_DEV
    if (ta.ablate & 8) {
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < RW; j++) acc ^= out[j];
        if (acc == 0x12345678u) a.dst[0] = 1;
        return true;
    }
#endif
    const int ncols = min(64, a.dw - x0);  // > 0
    if constexpr (FMT == 2) {
        // 16-bit BGR: three samples per pixel, lane = column (2-byte stores: a pixel is 6 bytes, rows are 2-byte aligned;
        // assembling the row into 4-byte stores through ds_bpermute was measured: no faster, 6 more instructions per pixel)
#pragma unroll
        for (int j = 0; j < RW; j++) {
            const int y = y0 + wave * RW + j;
            if (y >= a.dh) break;  // uniform
            if (col_live) {
                uint16_t *o = reinterpret_cast<uint16_t *>(a.dst + ((size_t)(uint32_t)y * a.pitch_dst + (uint32_t)x * 6u));
                o[0] = (uint16_t)(out[j] & 1023u), o[1] = (uint16_t)((out[j] >> 10) & 1023u), o[2] = (uint16_t)(out[j] >> 20);
            }
        }
    } else if constexpr (FMT == 3) {
        // P010 planes (the 10-bit path's encoder hand-off; definition: include/vstab.h, vstab_cvt_bgr16_p010): luma word for every
        // pixel, (U, V) words from the even-row / even-column pixels; four lanes' words are collected in the first lane of each
        // quad (DPP quad_perm) and stored as 8 bytes
#pragma unroll
        for (int j = 0; j < RW; j++) {
            const int y = y0 + wave * RW + j;
            if (y >= a.dh) break;  // uniform
            const int B = (int)(out[j] & 1023u), G = (int)((out[j] >> 10) & 1023u), R = (int)(out[j] >> 20);
            const uint32_t yb = (uint32_t)sat10((CRY * R + CGY * G + CBY * B + (1 << 19) + (64 << 20)) >> 20) << 6;
            const uint32_t y1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)yb, 0x55, 0xf, 0xf, true);
            const uint32_t y2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)yb, 0xaa, 0xf, 0xf, true);
            const uint32_t y3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)yb, 0xff, 0xf, 0xf, true);
            uint16_t *o = reinterpret_cast<uint16_t *>(a.dst + (size_t)(uint32_t)y * a.pitch_dst) + x;
            if (!(lane & 3) && col_live) {
                if (ta.dst_vec_ok && x + 4 <= a.dw) {
                    *reinterpret_cast<uint2 *>(o) = make_uint2(yb | (y1 << 16), y2 | (y3 << 16));
                } else {
                    const uint32_t w[4] = {yb, y1, y2, y3};
                    for (int i = 0; i < 4 && x + i < a.dw; i++) o[i] = (uint16_t)w[i];
                }
            }
            if (!(y & 1)) {
                const uint32_t U = (uint32_t)sat10((CRU * R + CGU * G + CBU * B + (1 << 19) + (512 << 20)) >> 20) << 6;
                const uint32_t V = (uint32_t)sat10((CBU * R + CGV * G + CBV * B + (1 << 19) + (512 << 20)) >> 20) << 6;
                const uint32_t cb = U | (V << 16);
                const uint32_t c2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)cb, 0xaa, 0xf, 0xf, true);
                uint16_t *c = reinterpret_cast<uint16_t *>(a.dst_uv + (size_t)(uint32_t)(y >> 1) * a.pitch_dst_uv) + x;
                if (!(lane & 3) && col_live) {
                    if (ta.dst_vec_ok && x + 2 < a.dw) {
                        *reinterpret_cast<uint2 *>(c) = make_uint2(cb, c2);
                    } else {
                        *reinterpret_cast<uint32_t *>(c) = cb;  // (U, V) pairs are 4-byte aligned (checked by the caller)
                        if (x + 2 < a.dw) *reinterpret_cast<uint32_t *>(c + 2) = c2;
                    }
                }
            }
        }
    } else if constexpr (FMT == 0) {
        const int p0 = (4 * lane) / 3, m3 = lane - 3 * (lane / 3);  // pixels p0, p0 + 1 feed dword `lane` (lane < 48)
        const uint32_t sel = m3 == 0 ? 0x04020100u : m3 == 1 ? 0x05040201u : 0x06050402u;
        const int nbytes = 3 * ncols, nfull = nbytes >> 2, rem = nbytes & 3;
        if (ta.dst_vec_ok && ncols == 64 && y0 + wave * RW + RW <= a.dh) {
            // the whole strip of this wave is inside the image (all but the last tile column / row): every cross-lane
            // read of the transposition first, then RW stores with nothing but a scalar address step between them
            uint32_t lo[RW], hi[RW];
#pragma unroll
            for (int j = 0; j < RW; j++) {
                lo[j] = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * p0, (int)out[j]);
                hi[j] = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * p0 + 4, (int)out[j]);
            }
            uint8_t *o = a.dst + ((size_t)(uint32_t)(y0 + wave * RW) * a.pitch_dst + (uint32_t)x0 * 3u);  // uniform
            if (lane < 48) {
#pragma unroll
                for (int j = 0; j < RW; j++) {
                    *reinterpret_cast<uint32_t *>(o + (uint32_t)(4 * lane)) = __builtin_amdgcn_perm(hi[j], lo[j], sel);
                    o += a.pitch_dst;
                }
            }
        } else
#pragma unroll
        for (int j = 0; j < RW; j++) {
            const int y = y0 + wave * RW + j;
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
        for (int j = 0; j < RW; j++) {
            const int y = y0 + wave * RW + j;
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
    VSTAB_STAMP(7);
    return true;
}

// Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8), so every XCD gets one contiguous band of
// output rows [band_y[k], band_y[k + 1]) in raster order: horizontally and vertically adjacent tiles then share an L2 and
// the 128-B lines their source and output rows straddle move once.  Inside its band an XCD first works through tiles of
// 4 RWB rows, then -- from row split_y[k] on -- through tiles of half that height (the last, partly filled round of
// workgroups then lasts half as long).  Placement and tile height only affect speed, never results.
// Register budgets.  The 64 x 16-tile kernels (small outputs, eight workgroups per CU) stay within 72 registers.  The 64 x 32-tile
// kernels get 80: at 72 the per-frame kernels kept ten scratch instructions (ragged-edge stores, the half-height tile's prologue) and
// the per-row kernels (nine matrix entries per row on top of everything else) spilled inside the map phase (4K, reference
// arithmetic, per row: 36.1 -> 33.2 us; per frame in the pipeline 26.8 -> 27.0 k frames/s, three alternating runs on one box).
// Their LDS holds them to four waves per SIMD anyway; 80 registers still leave room for a tracker wave (123) beside them.
template <int RWB, int MODE, int FMT, bool CACHED, int DEPTH = 8, int BLEND = 0>
__global__ void __launch_bounds__(256, DEPTH == 10 ? 5 : (RWB == 8 || map_mode_is_rs(MODE)) ? VSTAB_WARP_WAVES - 1 : VSTAB_WARP_WAVES) k_warp_fused(FusedArgs ta) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr int TH = 4 * RWB, TS = TH / 2;
    if (VSTAB_WARP_PRIO) __builtin_amdgcn_s_setprio(VSTAB_WARP_PRIO);
    const int k = (int)(blockIdx.x & 7u);
    const int y_lo = ta.band_y[k], y_sp = ta.split_y[k], y_hi = ta.band_y[k + 1];
    const int n_tall = ((y_sp - y_lo) / TH) * ta.tiles_x;  // y_sp - y_lo is a multiple of TH
    const int idx = (int)(blockIdx.x >> 3);
    int x0, ys, n_half;
    if (idx < n_tall) {
        const int row = idx / ta.tiles_x;
        x0 = (idx - row * ta.tiles_x) * 64, ys = y_lo + row * TH;
        // a tall tile whose box is over the LDS budget (the image centre, where the lens compresses most) is done as two
        // half-height tiles, one after the other
        n_half = warp_tile<RWB, RWB, MODE, FMT, CACHED, true, DEPTH, BLEND>(ta, smem, x0, ys) ? 0 : 2;
    } else {
        const int i2 = idx - n_tall, row = i2 / ta.tiles_x;
        x0 = (i2 - row * ta.tiles_x) * 64, ys = y_sp + row * TS, n_half = 1;
        if (ys >= y_hi) return;  // uniform for the workgroup (before any barrier)
    }
#pragma unroll 1
    for (int i = 0; i < n_half; i++) warp_tile<RWB, RWB / 2, MODE, FMT, CACHED, false, DEPTH, BLEND>(ta, smem, x0, ys + i * TS);
}

}  // namespace vstab

using namespace vstab;

// Launch of the fused kernel; called by warp_impl (vstab_warp.hip) after argument validation.
namespace vstab {
#ifdef VSTAB_DEV
static unsigned long long *g_dev_timing = nullptr;
extern "C" __attribute__((visibility("default"))) void vstab_dev_set_timing(void *p) { g_dev_timing = static_cast<unsigned long long *>(p); }
#endif

// The 10-bit pixel path on the same kernel (DEPTH 10): fisheye -> pinhole maps (modes 0 / 1 / 5, optionally a rotation per
// output row), both blends; called by vstab_warp_p010 when the planes allow 16-byte staging loads.
vstab_status launch_warp_fused10(const WarpArgs &a, const float params[17], int map_mode, int blend, const float *rot_bottom, bool p010_out, bool dst_vec_ok,
                                 hipStream_t st) {
    FusedArgs ta;
    ta.w = a;
    ta.p32 = {params[0] * 32.0f, params[1] * 32.0f, params[2] * 32.0f, params[3] * 32.0f, params[10], params[13], params[16]};
    ta.src_vec_ok = 1, ta.dst_vec_ok = dst_vec_ok;
    ta.qmap = nullptr, ta.qpitch = 0;
    for (int k = 0; k < 9; k++) ta.rs_d[k] = rot_bottom ? rot_bottom[k] - params[8 + k] : 0.0f;
    ta.rs_den = (float)(a.dh > 1 ? a.dh - 1 : 1);
#ifdef VSTAB_DEV
    ta.timing = nullptr, ta.ablate = 0, ta.lds_pad = 0;
#endif
    // 64 x 32 tiles, 40 KB of LDS.  The fp16 blend stages 8-byte pixels (three halves): half as many fit, and a tall tile whose box is
    // over that is done as two half-height tiles (SPLIT) -- measured faster than 64 x 16 tiles throughout (47.3 against 49.1 us at 4K).
    const int lds_kb = 40;
    const bool half = blend == VSTAB_BLEND_FP16;
    const int rwb = 8;
    const long tiles = (long)div_up(a.dw, 64) * div_up(a.dh, 4 * rwb);
    const dim3 grid(tile_schedule(ta, rwb, lds_kb, tiles > 1024 ? 0.5 : 0.0));
    if (half) ta.lds_capacity_px /= 2;
    const size_t lds_bytes = (size_t)lds_kb * 1024;
    const LaunchEvents ev = take_launch_events();
#define VSTAB_LAUNCH10(R, M, B, F)                                                                                                  \
    do {                                                                                                                            \
        if (ev.start) hipExtLaunchKernelGGL((k_warp_fused<R, M, F, false, 10, B>), grid, dim3(256), lds_bytes, st, ev.start, ev.stop, 0, ta); \
        else hipLaunchKernelGGL((k_warp_fused<R, M, F, false, 10, B>), grid, dim3(256), lds_bytes, st, ta);                            \
    } while (0)
#define VSTAB_LAUNCH10_M(M)                                                      \
    do {                                                                         \
        if (p010_out) {                                                          \
            if (half) VSTAB_LAUNCH10(8, M, VSTAB_BLEND_FP16, 3);                 \
            else VSTAB_LAUNCH10(8, M, VSTAB_BLEND_EXACT, 3);                     \
        } else if (half) VSTAB_LAUNCH10(8, M, VSTAB_BLEND_FP16, 2);              \
        else VSTAB_LAUNCH10(8, M, VSTAB_BLEND_EXACT, 2);                         \
    } while (0)
    if (rot_bottom) {
        if (map_mode == VSTAB_MAP_CREATEMAP_CL) VSTAB_LAUNCH10_M(MAP_RS_CREATEMAP_CL);
        else if (map_mode == VSTAB_MAP_CREATEMAP_CL_OPENCL) VSTAB_LAUNCH10_M(MAP_RS_CREATEMAP_CL_OPENCL);
        else VSTAB_LAUNCH10_M(MAP_RS_FISH_TO_RECT);
    } else {
        if (map_mode == VSTAB_MAP_CREATEMAP_CL) VSTAB_LAUNCH10_M(MAP_CREATEMAP_CL);
        else if (map_mode == VSTAB_MAP_CREATEMAP_CL_OPENCL) VSTAB_LAUNCH10_M(MAP_CREATEMAP_CL_OPENCL);
        else VSTAB_LAUNCH10_M(MAP_FISH_TO_RECT);
    }
#undef VSTAB_LAUNCH10_M
#undef VSTAB_LAUNCH10
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}

vstab_status launch_warp_fused(const WarpArgs &a, const float params[17], int map_mode, bool nv12_out, bool src_vec_ok, bool dst_vec_ok,
                               const void *qmap, int qpitch, const float *rot_bottom, hipStream_t st) {
    FusedArgs ta;
    ta.w = a;
    ta.p32 = {params[0] * 32.0f, params[1] * 32.0f, params[2] * 32.0f, params[3] * 32.0f, params[10], params[13], params[16]};
    ta.src_vec_ok = src_vec_ok, ta.dst_vec_ok = dst_vec_ok;
    ta.qmap = static_cast<const int2 *>(qmap), ta.qpitch = qpitch;
    for (int k = 0; k < 9; k++) ta.rs_d[k] = rot_bottom ? rot_bottom[k] - params[8 + k] : 0.0f;  // fp32, as the definition forms it
    ta.rs_den = (float)(a.dh > 1 ? a.dh - 1 : 1);
    if (rot_bottom)  // modes 0 / 1 / 5 only (checked by the caller)
        map_mode = map_mode == VSTAB_MAP_CREATEMAP_CL_OPENCL ? (int)MAP_RS_CREATEMAP_CL_OPENCL : map_mode + (int)MAP_RS_CREATEMAP_CL;
#ifdef VSTAB_DEV
    ta.timing = g_dev_timing;
    static const int ablate = getenv("VSTAB_ABLATE") ? atoi(getenv("VSTAB_ABLATE")) : 0;
    ta.ablate = ablate;
    ta.lds_pad = getenv("VSTAB_LDS_PAD") ? atoi(getenv("VSTAB_LDS_PAD")) & ~3 : 0;
#endif
    // Tile shape: 64 x 32 output pixels and 40 KB of LDS (4 workgroups per CU) when that gives the 1024 workgroup
    // slots of the chip a few rounds of tiles; 64 x 16 with 20 KB (8 per CU: 2048 slots, a 1080p output in ONE round)
    // for small outputs.
    const long tiles32 = (long)div_up(a.dw, 64) * div_up(a.dh, 32);
    int rwb = tiles32 < 1536 ? 4 : 8, lds_kb = rwb == 4 ? 20 : 40;
    // how many rounds of workgroup slots, counted from the end, use the half-height tiles: half a round when the launch
    // has several rounds (4K: 36.4 -> 35.8 us), none when everything is resident at once (1080p: 14.4 -> 13.6 us)
    double tail_rounds = (double)div_up(a.dw, 64) * div_up(a.dh, 4 * rwb) > 256.0 * (160 / lds_kb) ? 0.5 : 0.0;
#ifdef VSTAB_DEV
    if (const char *e = getenv("VSTAB_ROWS")) rwb = atoi(e) == 4 ? 4 : 8;
    if (const char *e = getenv("VSTAB_LDS_KB")) lds_kb = atoi(e);
    if (const char *e = getenv("VSTAB_TAIL_ROUNDS")) tail_rounds = atof(e);
#endif
    const size_t lds_bytes = (size_t)lds_kb * 1024;
    const dim3 grid(tile_schedule(ta, rwb, lds_kb, tail_rounds));
    // a profiling caller may have left an event pair for this launch: the kernel's own start / end stamps
    const LaunchEvents ev = take_launch_events();
#define VSTAB_LAUNCH(R, M, F, C)                                                                                            \
    do {                                                                                                                    \
        if (ev.start) hipExtLaunchKernelGGL((k_warp_fused<R, M, F, C>), grid, dim3(256), lds_bytes, st, ev.start, ev.stop, 0, ta); \
        else hipLaunchKernelGGL((k_warp_fused<R, M, F, C>), grid, dim3(256), lds_bytes, st, ta);                              \
    } while (0)
#define VSTAB_LAUNCH_RF(M, C)                                  \
    do {                                                       \
        if (rwb == 8 && !nv12_out) VSTAB_LAUNCH(8, M, 0, C);   \
        else if (rwb == 8) VSTAB_LAUNCH(8, M, 1, C);           \
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
            case MAP_RS_CREATEMAP_CL_OPENCL: VSTAB_LAUNCH_RF(MAP_RS_CREATEMAP_CL_OPENCL, false); break;
            default: VSTAB_LAUNCH_RF(MAP_RS_FISH_TO_RECT, false); break;
        }
    }
#undef VSTAB_LAUNCH_RF
#undef VSTAB_LAUNCH
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}


// Kernels of this translation unit are one code object, loaded by the runtime at the first launch of any of them.  Touching one of them
// here (vstab_preload_kernels) moves that load to a moment the caller chooses.
vstab_status preload_fused_kernels() {
    hipFuncAttributes at;
    VSTAB_HIP_TRY(hipFuncGetAttributes(&at, reinterpret_cast<const void *>((&k_warp_fused<8, MAP_CREATEMAP_CL_OPENCL, 0, false, 8, 0>))));
    return VSTAB_OK;
}

}  // namespace vstab
