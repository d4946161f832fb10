// vstab_warp_planar.hip -- the PLANE-WISE undistort-remap kernel (gfx950 / MI355X): NV12 -> NV12 and P010 -> P010 with no colour
// round trip (SURVEY.md 8(f) row 2 as it is written; the route the CLI's filter and encoders use: render.ts:606-607, 664-665, 688,
// 275-281).  Definition (include/vstab.h, VSTAB_OUT_NV12_PLANAR; restated in plain C in the test infrastructure):
//   luma    cv::remap(INTER_LINEAR, BORDER_CONSTANT 16) of the Y plane with the map and quantisation of the BGR path
//           (createMap.cl:13-50, FrameSourceWarp.cpp:306-312);
//   chroma  cv::remap(INTER_LINEAR, BORDER_CONSTANT (128, 128)) of the interleaved UV plane; chroma sample (cx, cy) takes the map
//           entry of luma pixel (2 cx, 2 cy), halved (exact) and quantised again: cvRound(32 * (map / 2)).
// The kernel is k_warp_fused's structure (vstab_warp_fused.hip) without its colour conversion: probe -> load -> map -> stage -> sample
// -> store, one workgroup per 64-column tile, lane = output column, a wave owns RW consecutive rows.  What differs:
//   stage   the box is staged AS IT IS -- luma samples and chroma pairs, 1.5 samples per source pixel instead of a 4-byte BGRx pixel
//           -- by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write); chunks outside the source are written as
//           limited-range black;
//   sample  a luma pixel is four byte reads (the two taps of two rows, each at its natural alignment), two v_dot4_u32_u8 and one
//           v_dot2_u32_u16; a thread's eight luma pixels come with TWO chroma pixels (the even-row map entries of the even lanes, dealt
//           to lane pairs with DPP), each four 16-bit reads and six dot products;
//   store   result bytes are transposed through 768 bytes of LDS per wave, so every lane stores 8 (luma) + 4 (chroma) contiguous bytes.
// 10-bit samples (DEPTH 10): the box holds the P010 WORDS as they came; a pair of taps becomes two ten-bit values with one packed shift
// (v_pk_lshrrev_b16), the exact blend is v_dot2_u32_u16 in place of v_dot4, the binary16 blend is packed two samples (or U and V) wide.
// Bit-exactness never depends on the box: a footprint outside it is sampled from global memory with the same integers.
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

// LDS reads are issued at their NATURAL alignment only: gfx950 executes an unaligned ds_read_u16 / ds_read_b32 lane by lane -- 64 cycles
// of the CU's LDS pipe against 2.3 (tools/probe_lds.hip, profiles/r05_lds_access_cost.txt) -- so the two taps of a row, which start at
// any byte, are two byte reads (two 16-bit reads for 16-bit samples and for chroma byte pairs) joined by one v_lshl_or_b32.
typedef __attribute__((address_space(3))) uint8_t LdsU8;
typedef __attribute__((address_space(3))) uint16_t LdsU16;
typedef __attribute__((address_space(3))) uint32_t LdsU32;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

template <int DEPTH>
struct Planar {
    static constexpr int BPS = DEPTH == 10 ? 2 : 1;            // bytes per sample
    static constexpr int BW = 16 / BPS;                        // luma pixels of a staging block row (one 16-byte load)
    static constexpr int BLACK_Y = DEPTH == 10 ? 64 : 16;      // limited-range black: what cv::remap's Scalar(0) border of the BGR path is here
    static constexpr int BLACK_C = DEPTH == 10 ? 512 : 128;
    static constexpr uint32_t BLACK_Y_DWORD = DEPTH == 10 ? 0x10001000u : 0x10101010u;  // as staged in LDS (10 bits: P010 words, value << 6)
    static constexpr uint32_t BLACK_C_DWORD = DEPTH == 10 ? 0x80008000u : 0x80808080u;
    static constexpr int SCRATCH_PER_WAVE = 768 * BPS;         // 8 rows x 64 luma + 4 rows x 32 chroma pairs
};

// ---- the rare path: one sample straight from global memory, taps outside the source = the border value ---------------------------
template <int DEPTH>
__device__ __forceinline__ int plane_tap(const uint8_t *plane, size_t pitch, int X, int Y, int cn, int c, int w, int h, int border) {
    if ((unsigned)X >= (unsigned)w || (unsigned)Y >= (unsigned)h) return border;
    const uint8_t *p = plane + (size_t)Y * pitch + ((size_t)X * cn + c) * Planar<DEPTH>::BPS;
    if constexpr (DEPTH == 10) return (int)(*reinterpret_cast<const uint16_t *>(p) >> 6);
    else return (int)*p;
}
template <int DEPTH, int BLEND>
__device__ __forceinline__ int blend4(int p00, int p01, int p10, int p11, int fx, int fy) {
    const int w00 = (32 - fx) * (32 - fy), w01 = fx * (32 - fy), w10 = (32 - fx) * fy, w11 = fx * fy;
    if constexpr (DEPTH == 10 && BLEND == VSTAB_BLEND_FP16) return blend_fp16(p00, p01, p10, p11, w00, w01, w10, w11);
    else return (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + 512) >> 10;
}
// sx, sy = rint(32 * position in the plane); cn interleaved channels of a w x h plane
template <int DEPTH, int BLEND>
__device__ __forceinline__ int gather_sample(const uint8_t *plane, size_t pitch, int w, int h, int cn, int c, int sx, int sy, int border) {
    const int X = sx >> 5, Y = sy >> 5;
    if (X >= w || X + 1 < 0 || Y >= h || Y + 1 < 0) return border;
    return blend4<DEPTH, BLEND>(plane_tap<DEPTH>(plane, pitch, X, Y, cn, c, w, h, border), plane_tap<DEPTH>(plane, pitch, X + 1, Y, cn, c, w, h, border),
                                plane_tap<DEPTH>(plane, pitch, X, Y + 1, cn, c, w, h, border), plane_tap<DEPTH>(plane, pitch, X + 1, Y + 1, cn, c, w, h, border),
                                sx & 31, sy & 31);
}

// (h0 * (32 - fy) + h1 * fy + 512) >> 10 from the two rows' horizontal sums.  Plain C++ on purpose: h0 / h1 come out of v_dot4 / v_dot2,
// whose results the next few VALU instructions must not read (the compiler pads ITS instructions; it cannot see into inline assembly --
// hand-written consumers here read stale sums).
__device__ __forceinline__ int lerp_rows(uint32_t h0, uint32_t h1, uint32_t fy) {
    const uint32_t base = (h0 << 5) + 512u;
    return (int)((base + (uint32_t)__mul24((int)(h1 - h0), (int)fy)) >> 10);
}

// The 8-bit blend with its result left at bits 16..23: 64 * ((h0 (32 - fy) + h1 fy + 512)) = dot2((4 h0, 4 h1), (16 (32 - fy), 16 fy)) + 32768
// where 4 h = dot4(taps of a row, (4 (32 - fx), 4 fx)) (<= 32640: a 16-bit field); every factor 4 / 16 is a shift, so the byte is the
// integer cv::remap computes.  wx4 = the horizontal weights at the bytes of the two taps.
__device__ __forceinline__ uint32_t blend8_scaled(uint32_t t_top, uint32_t t_bot, uint32_t wx4, uint32_t fy) {
    const uint32_t h0 = __builtin_amdgcn_udot4(t_top, wx4, 0u, false), h1 = __builtin_amdgcn_udot4(t_bot, wx4, 0u, false);
    const uint32_t wy16 = fy * 1048560u + 512u;  // 16 (32 - fy) | 16 fy << 16
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, h0 | (h1 << 16)), __builtin_bit_cast(u16x2, wy16), 32768u, false);
}

// ---- 16-bit samples.  Two P010 words -> their two ten-bit values (one v_pk_lshrrev_b16) --------------------------------------------
__device__ __forceinline__ uint32_t sig10_x2(uint32_t words) {
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, words) >> (u16x2){6, 6}));
}
// two ten-bit values -> two binary16 numbers: the pattern 0x6400 | v is 1024 + v
__device__ __forceinline__ half2v half_of10_x2(uint32_t v) {
    return __builtin_bit_cast(half2v, v | 0x64006400u) - (half2v){(_Float16)1024.0f, (_Float16)1024.0f};
}
// two five-bit fractions f -> f / 32 in binary16: (1024 + f) / 32 - 32 in one fused multiply-add (32 + f / 32 has ulp 1 / 32: exact)
__device__ __forceinline__ half2v frac_half_x2(uint32_t f2) {
    return __builtin_elementwise_fma(__builtin_bit_cast(half2v, f2 | 0x64006400u), (half2v){(_Float16)0.03125f, (_Float16)0.03125f},
                                     (half2v){(_Float16)-32.0f, (_Float16)-32.0f});
}
// VSTAB_BLEND_FP16 (vstab_device10.hpp blend_fp16: four fused multiply-adds in the order 00, 01, 10, 11 with weights w / 1024 -- products
// of multiples of 1 / 32, exact in binary16 --, round to nearest even, clamp) on TWO independent samples at once, one per half; taps and
// fractions as binary16 pairs.  Returns the two ten-bit values (value | value << 16); as blend_bgr10h does it for B and G.
__device__ __forceinline__ uint32_t blend10h_x2(half2v p00, half2v p01, half2v p10, half2v p11, half2v fx, half2v fy) {
    const half2v one = {(_Float16)1.0f, (_Float16)1.0f};
    const half2v gx = one - fx, gy = one - fy;
    half2v acc = p00 * (gx * gy);  // fma(p, w, 0)
    acc = __builtin_elementwise_fma(p01, fx * gy, acc);
    acc = __builtin_elementwise_fma(p10, gx * fy, acc);
    acc = __builtin_elementwise_fma(p11, fx * fy, acc);
    const half2v top = {(_Float16)1023.0f, (_Float16)1023.0f}, magic = {(_Float16)1024.0f, (_Float16)1024.0f};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(acc, top) + magic) & 0x03ff03ffu;  // ulp 1 at 1024: the adder rounds
}

// NB bytes of this lane from the wave's transposition scratch to global memory (NB = 1, 2, 4, 8, 16)
// (the scratch is written as samples and read as vectors: may_alias types, so that the reads stay behind the writes)
typedef uint16_t __attribute__((may_alias)) u16_alias;
typedef uint32_t __attribute__((may_alias)) u32_alias;
typedef uint32_t u32x2_alias __attribute__((ext_vector_type(2), may_alias));
typedef uint32_t u32x4_alias __attribute__((ext_vector_type(4), may_alias));
template <int NB>
__device__ __forceinline__ void scratch_to_global(const uint8_t *lds, uint8_t *dst) {
    if constexpr (NB == 1) *dst = *lds;
    else if constexpr (NB == 16) *reinterpret_cast<u32x4_alias *>(dst) = *reinterpret_cast<const u32x4_alias *>(lds);
    else if constexpr (NB == 8) *reinterpret_cast<u32x2_alias *>(dst) = *reinterpret_cast<const u32x2_alias *>(lds);
    else if constexpr (NB == 4) *reinterpret_cast<u32_alias *>(dst) = *reinterpret_cast<const u32_alias *>(lds);
    else *reinterpret_cast<u16_alias *>(dst) = *reinterpret_cast<const u16_alias *>(lds);
}

// ---------------------------------------------------------------------------------------------------------------------
// One output tile: 64 luma columns x 4 RW rows at (x0, y0) and the 32 x 2 RW chroma pairs under them.  Returns false -- right
// behind the probe -- when SPLIT is set and the box is over the LDS budget (the caller then does two half-height tiles).
// LDS: [0, 32) tile header | 4 x SCRATCH_PER_WAVE transposition scratch | luma box (hb rows of pw bytes) | chroma box (hb / 2 rows of pw bytes)
// ---------------------------------------------------------------------------------------------------------------------
template <int RWB, int RW, int MODE, bool SPLIT, int DEPTH, int BLEND>
__device__ __forceinline__ bool warp_tile_planar(const FusedArgs &ta, uint32_t *smem, const int x0, const int y0) {
    using P = Planar<DEPTH>;
    constexpr int BPS = P::BPS, BW = P::BW;  // a staging chunk: BW samples = 16 bytes
    constexpr int TH = 4 * RW;
    constexpr int STAGE_MAX = 1 << 12;  // (the probe's bound on staging blocks per thread belongs to register staging: LDS-DMA takes any number of row groups)
    constexpr int CR = RW / 2;        // chroma rows of a wave
    constexpr int NS = (RW + 3) / 4;  // chroma pixels per thread: CR rows x 32 columns over 64 lanes (RW = 2: the even lanes only)
    constexpr int QB = QMAGIC_BITS;
    uint8_t *const scratch = reinterpret_cast<uint8_t *>(smem) + 32;
    uint8_t *const tile = scratch + 4 * P::SCRATCH_PER_WAVE;
    const WarpArgs &a = ta.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float rfx = rcp_refined(a.p.ofx), rfy = rcp_refined(a.p.ofy);

    // ---- probe (wave 0; the other waves wait at the barrier.  Taking the probing wave in turn by workgroup -- in case the waves of
    // a workgroup always landed on the same SIMDs -- changes nothing: 26.2 against 26.3 us at 4K) -------------------------------------
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        probe_tile<TH, STAGE_MAX, MODE, false, true, BW>(ta, x0, y0, lane, rfx, rfy, smem);
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
    const int pw = wb * BPS;                       // LDS row pitch in bytes, of both boxes (a chroma pair takes as much as two luma samples)
    const int luma_bytes = pw * hb;

    // ---- load: the box goes from global memory straight into LDS (LDS-DMA, global_load_lds_dwordx4: no staging registers, no
    // ds_write).  One wave-instruction writes 64 consecutive 16-byte chunks of LDS from 64 addresses of its lanes' choosing; the box is rows
    // of ux_n chunks, so lane i of an instruction takes chunk (i / ux_n, i % ux_n) of a group of 64 / ux_n rows and the LDS image is the box,
    // row after row.  The lane's part of the source offset is computed ONCE per tile; an instruction adds a scalar row offset.  The four
    // waves take the row groups of the luma box, then of the chroma box, in turn.  Chunks outside the source (tiles at the frame's edge)
    // are not fetched but written as limited-range black. --------------------------------------------------------------------------------
    const int ux_n = wb / BW;
    {
        if (use_lds) {
            const float rn = __builtin_amdgcn_rcpf((float)ux_n);
            int r0 = (int)((float)lane * rn), col = lane - r0 * ux_n;  // divmod(lane, ux_n): the float quotient is off by at most one
            if (col < 0) col += ux_n, r0--;
            if (col >= ux_n) col -= ux_n, r0++;
            const int rpi_f = __builtin_amdgcn_readfirstlane((int)(64.0f * rn + 0.001f));  // rows per instruction: floor(64 / ux_n)
            const int rpi = rpi_f * ux_n <= 64 ? rpi_f : rpi_f - 1;
            const bool active = r0 < rpi;
            const int colp = bx0 + BW * col;  // first source sample of the lane's chunk within a row (a chroma row: the same bytes per luma column)
            const int sw_al = a.sw & ~(BW - 1);
            const bool col_ok = (uint32_t)colp < (uint32_t)sw_al;
            const uint32_t pitch_y = (uint32_t)a.pitch_y, pitch_uv = (uint32_t)a.pitch_uv;  // < 2^24, frame < 4 GiB (host check)
            const uint32_t lane_y = (uint32_t)r0 * pitch_y + (uint32_t)(colp * BPS), lane_c = (uint32_t)r0 * pitch_uv + (uint32_t)(colp * BPS);
            const int hc = hb >> 1;
            const int nl = (hb + rpi - 1) / rpi, nc = (hc + rpi - 1) / rpi;
            const bool interior = bx0 >= 0 && by0 >= 0 && bx0 + wb <= sw_al && by0 + hb <= a.sh;  // uniform: no chunk outside the source
#pragma unroll 1
            for (int t = wave; t < nl + nc; t += 4) {
                const bool luma = t < nl;                      // uniform
                const int k = (luma ? t : t - nl) * rpi;       // first box row of the group
                const int R = k + r0;                          // this lane's box row
                const int srow = (luma ? by0 : by0 >> 1) + k;  // first source row of the group (may be negative at the frame's edge)
                bool ok = active && R < (luma ? hb : hc), fill = false;
                if (!interior) {
                    const bool in = col_ok && (uint32_t)(srow + r0) < (uint32_t)(luma ? a.sh : a.sh >> 1);
                    fill = ok && !in, ok = ok && in;
                }
                uint8_t *const ldst = tile + (luma ? 0 : luma_bytes) + k * pw;  // uniform; the lanes' chunks follow each other from here
#ifdef VSTAB_DEV
                if (ta.ablate & 16) ok = false;  // timing only: nothing is staged
#endif
                if (ok) {
                    const uint32_t off = (uint32_t)srow * (luma ? pitch_y : pitch_uv) + (luma ? lane_y : lane_c);  // modulo 2^32: a negative srow is made up for by r0
                    __builtin_amdgcn_global_load_lds((luma ? a.y : a.uv) + (size_t)off, (__attribute__((address_space(3))) void *)ldst, 16, 0, 0);
                }
                if (fill) {
                    const uint32_t kb = luma ? P::BLACK_Y_DWORD : P::BLACK_C_DWORD;
                    *reinterpret_cast<uint4 *>(ldst + lane * 16) = make_uint4(kb, kb, kb, kb);
                }
            }
        }
    }
    // ---- map: RW exact evaluations per thread, and the chroma positions of the even rows (vstab_warp_tile.hpp).  The rounding constants
    // carry the origin of the staged box (FOLD): the registers hold positions RELATIVE to the box, in 1/32 pixel -------------------------
    int qxb[RW], qyb[RW], qcx[RW / 2], qcy[RW / 2];
    {
        const float qm[4] = {QMAGIC - (float)(32 * bx0), QMAGIC - (float)(32 * by0), QMAGIC - (float)(32 * (bx0 >> 1)), QMAGIC - (float)(32 * (by0 >> 1))};
        map_phase<RW, MODE, false, true, true>(ta, x, y0, wave, lane, rfx, rfy, qxb, qyb, qcx, qcy, qm);
    }
#pragma unroll
    for (int j = 0; j < RW; j++) asm volatile("" : "+v"(qxb[j]), "+v"(qyb[j]) : : "memory");
#pragma unroll
    for (int j = 0; j < RW / 2; j++) asm volatile("" : "+v"(qcx[j]), "+v"(qcy[j]));

    __syncthreads();

    // ---- sample + blend: luma.  Results leave this block SCALED: out[j] = 64 * (value of the blend) + 32768 for 8-bit samples (the byte is
    // bits 16..23: ds_write_b8_d16_hi stores it without a shift); 16-bit samples: outw[j / 2] = the P010 words of rows j and j + 1 ------
    const bool col_live = x < a.dw;
    constexpr int OSH = DEPTH == 10 ? 0 : 16;   // where the result sits in out[] / cu[] / cv[]
    int out[RW];
    uint32_t outw[(RW + 1) / 2];                // (16-bit samples)
    {
        // every footprint of the wave inside the staged box?  relative positions: 0 <= X and X + 1 <= wb - 1, i.e. QB <= q < QB + 32 * (wb - 1)
        int mnx = qxb[0], mxx = qxb[0], mny = qyb[0], mxy = qyb[0];
#pragma unroll
        for (int j = 1; j < RW; j++) mnx = min(mnx, qxb[j]), mxx = max(mxx, qxb[j]), mny = min(mny, qyb[j]), mxy = max(mxy, qyb[j]);
        const int hix = QB + 32 * (wb - 1), hiy = QB + 32 * (hb - 1);
        const bool outside_box = !use_lds || mnx < QB || mxx >= hix || mny < QB || mxy >= hiy;
#ifdef VSTAB_DEV
        if (ta.ablate & 2) {  // timing only: no taps, no blend
#pragma unroll
            for (int j = 0; j < RW; j++) out[j] = qxb[j] ^ qyb[j], outw[j / 2] = (uint32_t)(qxb[j] ^ qyb[j]);
        } else
#endif
        if (!__builtin_amdgcn_ballot_w64(outside_box)) {
            // tap address = tile + Y * pw + X * BPS with X = bits 5..21 of q (QB has none of them set, and X < 2^17): two bit-field
            // extracts and one multiply-add; the tile's LDS address is a link-time constant and rides in the reads' offset field
            constexpr int TG = RW < 4 ? RW : 4;  // rows whose tap reads are issued before the first blend
#pragma unroll
            for (int j0 = 0; j0 < RW; j0 += TG) {
                uint32_t t0[TG], t1[TG];
#pragma unroll
                for (int j = 0; j < TG; j++) {
                    const uint32_t xr = __builtin_amdgcn_ubfe((uint32_t)qxb[j0 + j], 5, 17), yr = __builtin_amdgcn_ubfe((uint32_t)qyb[j0 + j], 5, 17);
                    uint32_t ad;
                    // t = (tap X) | (tap X + 1) << 16 of the upper and of the lower row
                    if constexpr (DEPTH == 10) {
                        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(ad) : "v"(yr), "s"(pw), "v"(xr << 1));
                        // (volatile: left alone, the compiler joins the two reads of a row into ONE ds_read_b32 at a 2-byte boundary -- 64 cycles)
                        const volatile LdsU16 *u = reinterpret_cast<const volatile LdsU16 *>((__attribute__((address_space(3))) uint8_t *)tile + ad);
                        const volatile LdsU16 *l = reinterpret_cast<const volatile LdsU16 *>((__attribute__((address_space(3))) uint8_t *)tile + (ad + (uint32_t)pw));
                        t0[j] = (uint32_t)u[0] | ((uint32_t)u[1] << 16), t1[j] = (uint32_t)l[0] | ((uint32_t)l[1] << 16);
                    } else {
                        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(ad) : "v"(yr), "s"(pw), "v"(xr));
                        const LdsU8 *u = (__attribute__((address_space(3))) uint8_t *)tile + ad, *l = (__attribute__((address_space(3))) uint8_t *)tile + (ad + (uint32_t)pw);
                        t0[j] = (uint32_t)u[0] | ((uint32_t)u[1] << 16), t1[j] = (uint32_t)l[0] | ((uint32_t)l[1] << 16);
                    }
                }
                if constexpr (DEPTH == 10) {
                    // rows j and j + 1 together: their words leave as one register
#pragma unroll
                    for (int j = 0; j < TG; j += 2) {
                        const uint32_t qa = (uint32_t)qxb[j0 + j], qb = (uint32_t)qxb[j0 + j + 1], ra = (uint32_t)qyb[j0 + j], rb = (uint32_t)qyb[j0 + j + 1];
                        uint32_t v2;
                        if constexpr (BLEND == VSTAB_BLEND_FP16) {
                            // tap 00 of both samples in one register, tap 01 in the next, ..: the blend runs two samples wide
                            const uint32_t p00 = sig10_x2(__builtin_amdgcn_perm(t0[j + 1], t0[j], 0x05040100u)), p01 = sig10_x2(__builtin_amdgcn_perm(t0[j + 1], t0[j], 0x07060302u));
                            const uint32_t p10 = sig10_x2(__builtin_amdgcn_perm(t1[j + 1], t1[j], 0x05040100u)), p11 = sig10_x2(__builtin_amdgcn_perm(t1[j + 1], t1[j], 0x07060302u));
                            const uint32_t fx2 = __builtin_amdgcn_perm(qb, qa, 0x05040100u) & 0x001f001fu, fy2 = __builtin_amdgcn_perm(rb, ra, 0x05040100u) & 0x001f001fu;
                            v2 = blend10h_x2(half_of10_x2(p00), half_of10_x2(p01), half_of10_x2(p10), half_of10_x2(p11), frac_half_x2(fx2), frac_half_x2(fy2));
                        } else {
                            uint32_t v[2];
#pragma unroll
                            for (int k = 0; k < 2; k++) {
                                const uint32_t fx = (k ? qb : qa) & 31u, fy = (k ? rb : ra) & 31u;
                                const uint32_t wx = fx * 65535u + 32u;  // (32 - fx) | fx << 16
                                const uint32_t h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, sig10_x2(t0[j + k])), __builtin_bit_cast(u16x2, wx), 0u, false);
                                const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, sig10_x2(t1[j + k])), __builtin_bit_cast(u16x2, wx), 0u, false);
                                v[k] = (uint32_t)lerp_rows(h0, h1, fy);
                            }
                            v2 = v[0] | (v[1] << 16);
                        }
                        outw[(j0 + j) / 2] = v2 << 6;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < TG; j++) {
                        const uint32_t fx = (uint32_t)qxb[j0 + j] & 31u, fy = (uint32_t)qyb[j0 + j] & 31u;
                        out[j0 + j] = (int)blend8_scaled(t0[j], t1[j], fx * 262140u + 128u, fy);  // 4 (32 - fx) at byte 0, 4 fx at byte 2
                    }
                }
            }
        } else {
            // rare (a box that had to be cut, degenerate rotation, box over the LDS budget, unaligned planes): straight from global memory
#pragma unroll 1
            for (int j = 0; j < RW; j++) {
                int sx = qxb[0], sy = qyb[0];
#pragma unroll
                for (int k = 1; k < RW; k++) sx = j == k ? qxb[k] : sx, sy = j == k ? qyb[k] : sy;
                const int v = gather_sample<DEPTH, BLEND>(a.y, a.pitch_y, a.sw, a.sh, 1, 0, sx - QB + 32 * bx0, sy - QB + 32 * by0, P::BLACK_Y) << OSH;
#pragma unroll
                for (int k = 0; k < RW; k++) out[k] = j == k ? v : out[k];
            }
            if constexpr (DEPTH == 10) {
#pragma unroll
                for (int k = 0; k < RW; k += 2) outw[k / 2] = ((uint32_t)out[k] | ((uint32_t)out[k + 1] << 16)) << 6;
            }
        }
    }

    // ---- sample + blend: chroma.  Lane pair (2k, 2k + 1) shares chroma column k; the even lane holds its map entries for the even
    // rows j = 0, 2, ..: the even lane keeps rows j = 4 s, the odd lane takes rows j = 4 s + 2 (chroma rows 2 s and 2 s + 1) -------------
    int cu[NS], cv[NS];
    uint32_t cw[NS];  // (16-bit samples) the pair's P010 words, U | V << 16
    {
        int ccx[NS], ccy[NS];
        const bool odd = lane & 1;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int so = 2 * s + 1 < CR ? 2 * s + 1 : 2 * s;  // (RW = 2: one chroma row; the odd lanes shadow their neighbour and store nothing)
            const int ox = __builtin_amdgcn_mov_dpp(qcx[so], 0xa0, 0xf, 0xf, true), oy = __builtin_amdgcn_mov_dpp(qcy[so], 0xa0, 0xf, 0xf, true);  // quad_perm [0,0,2,2]
            ccx[s] = odd ? ox : qcx[2 * s], ccy[s] = odd ? oy : qcy[2 * s];
        }
        int mnx = ccx[0], mxx = ccx[0], mny = ccy[0], mxy = ccy[0];
#pragma unroll
        for (int s = 1; s < NS; s++) mnx = min(mnx, ccx[s]), mxx = max(mxx, ccx[s]), mny = min(mny, ccy[s]), mxy = max(mxy, ccy[s]);
        const int cbx0 = bx0 >> 1, cby0 = by0 >> 1, cwb = wb >> 1, chb = hb >> 1;  // the chroma box, in chroma pixels (bx0, by0 even, also when negative)
        const int hix = QB + 32 * (cwb - 1), hiy = QB + 32 * (chb - 1);
        const bool outside_box = !use_lds || mnx < QB || mxx >= hix || mny < QB || mxy >= hiy;
#ifdef VSTAB_DEV
        if (ta.ablate & 4) {  // timing only: no chroma taps, no blend
#pragma unroll
            for (int s = 0; s < NS; s++) cu[s] = ccx[s], cv[s] = ccy[s], cw[s] = (uint32_t)(ccx[s] ^ ccy[s]);
        } else
#endif
        if (!__builtin_amdgcn_ballot_w64(outside_box)) {
            constexpr uint32_t CPX = 2 * BPS;  // bytes of a chroma pair
            const __attribute__((address_space(3))) uint8_t *const ctile = (__attribute__((address_space(3))) uint8_t *)tile + luma_bytes;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const uint32_t xr = __builtin_amdgcn_ubfe((uint32_t)ccx[s], 5, 17), yr = __builtin_amdgcn_ubfe((uint32_t)ccy[s], 5, 17);
                const uint32_t fx = (uint32_t)ccx[s] & 31u, fy = (uint32_t)ccy[s] & 31u;
                uint32_t ad;
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(ad) : "v"(yr), "s"(pw), "v"(xr * CPX));
                if constexpr (DEPTH == 10) {
                    // (U0 | V0 << 16, U1 | V1 << 16) of both rows
                    const LdsU32 *pt = reinterpret_cast<const LdsU32 *>(ctile + ad), *pb = reinterpret_cast<const LdsU32 *>(ctile + (ad + (uint32_t)pw));
                    const uint32_t a0 = sig10_x2(pt[0]), a1 = sig10_x2(pt[1]), b0 = sig10_x2(pb[0]), b1 = sig10_x2(pb[1]);
                    uint32_t v2;
                    if constexpr (BLEND == VSTAB_BLEND_FP16) {
                        // U and V side by side as they lie in memory: the blend runs on the pair
                        v2 = blend10h_x2(half_of10_x2(a0), half_of10_x2(a1), half_of10_x2(b0), half_of10_x2(b1), frac_half_x2(fx * 0x10001u), frac_half_x2(fy * 0x10001u));
                    } else {
                        const u16x2 wx = __builtin_bit_cast(u16x2, fx * 65535u + 32u);  // (32 - fx) | fx << 16
                        auto row = [&](uint32_t l, uint32_t r, uint32_t sel) { return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(r, l, sel)), wx, 0u, false); };
                        const uint32_t u = (uint32_t)lerp_rows(row(a0, a1, 0x05040100u), row(b0, b1, 0x05040100u), fy);
                        const uint32_t v = (uint32_t)lerp_rows(row(a0, a1, 0x07060302u), row(b0, b1, 0x07060302u), fy);
                        v2 = u | (v << 16);
                    }
                    cw[s] = v2 << 6;
                } else {
                    const LdsU16 *pt = reinterpret_cast<const LdsU16 *>(ctile + ad), *pb = reinterpret_cast<const LdsU16 *>(ctile + (ad + (uint32_t)pw));
                    const uint32_t tt = (uint32_t)pt[0] | ((uint32_t)pt[1] << 16), tb = (uint32_t)pb[0] | ((uint32_t)pb[1] << 16);  // bytes U0 V0 U1 V1
                    const uint32_t wu = fx * 262140u + 128u;  // 4 (32 - fx) at byte 0, 4 fx at byte 2: against U; << 8: against V
                    cu[s] = (int)blend8_scaled(tt, tb, wu, fy), cv[s] = (int)blend8_scaled(tt, tb, wu << 8, fy);
                }
            }
        } else {
#pragma unroll 1
            for (int s = 0; s < NS; s++) {
                int sx = ccx[0], sy = ccy[0];
#pragma unroll
                for (int k = 1; k < NS; k++) sx = s == k ? ccx[k] : sx, sy = s == k ? ccy[k] : sy;
                const int u = gather_sample<DEPTH, BLEND>(a.uv, a.pitch_uv, a.sw >> 1, a.sh >> 1, 2, 0, sx - QB + 32 * cbx0, sy - QB + 32 * cby0, P::BLACK_C) << OSH;
                const int v = gather_sample<DEPTH, BLEND>(a.uv, a.pitch_uv, a.sw >> 1, a.sh >> 1, 2, 1, sx - QB + 32 * cbx0, sy - QB + 32 * cby0, P::BLACK_C) << OSH;
#pragma unroll
                for (int k = 0; k < NS; k++) cu[k] = s == k ? u : cu[k], cv[k] = s == k ? v : cv[k];
            }
            if constexpr (DEPTH == 10) {
#pragma unroll
                for (int k = 0; k < NS; k++) cw[k] = ((uint32_t)cu[k] | ((uint32_t)cv[k] << 16)) << 6;
            }
        }
    }

    // ---- store: through the wave's scratch, so that a lane stores contiguous bytes of ONE row ------------------------------
#ifdef VSTAB_DEV
    if (ta.ablate & 8) {  // timing only: no stores
        int acc = 0;
#pragma unroll
        for (int j = 0; j < RW; j++) acc ^= DEPTH == 10 ? (int)outw[j / 2] : out[j];
#pragma unroll
        for (int s = 0; s < NS; s++) acc ^= DEPTH == 10 ? (int)cw[s] : cu[s] ^ cv[s];
        if (acc == 0x12345678) a.dst[0] = 1;
        return true;
    }
#endif
    const int yw = y0 + wave * RW;                      // first luma row of this wave (even)
    const int ncols = min(64, a.dw - x0);               // > 0
    if (ta.dst_vec_ok && ncols == 64 && yw + RW <= a.dh) {
        uint8_t *const scr = scratch + wave * P::SCRATCH_PER_WAVE;
        uint8_t *const scr_c = scr + RW * 64 * BPS;
        {
            uint8_t *w = scr + lane * BPS;
#pragma unroll
            for (int j = 0; j < RW; j++) {
                if constexpr (DEPTH == 10) *reinterpret_cast<u16_alias *>(w + j * 128) = (uint16_t)(j & 1 ? outw[j / 2] >> 16 : outw[j / 2]);  // (ds_write_b16, _d16_hi)
                else w[j * 64] = (uint8_t)((uint32_t)out[j] >> 16);  // (ds_write_b8_d16_hi)
            }
            uint8_t *wc = scr_c + (lane & 1) * 64 * BPS + (lane >> 1) * 2 * BPS;  // chroma row (lane & 1) + 2 s, pair lane >> 1
            if (CR > 1 || !(lane & 1)) {
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    if constexpr (DEPTH == 10) *reinterpret_cast<u32_alias *>(wc + s * 256) = cw[s];
                    else *reinterpret_cast<u16_alias *>(wc + s * 128) = (uint16_t)__builtin_amdgcn_perm((uint32_t)cv[s], (uint32_t)cu[s], 0x0c0c0602u);  // U | V << 8 from bits 16..23
                }
            }
        }
        // luma: RW rows of 64 * BPS bytes = 64 lanes x RW * BPS bytes; chroma: RW / 2 rows of 64 * BPS bytes = 64 lanes x RW * BPS / 2
        constexpr int NBY = RW * BPS, NBC = RW * BPS / 2;
        constexpr int LPR_Y = 64 * BPS / NBY, LPR_C = 64 * BPS / NBC;  // lanes per row
        {
            const int r = lane / LPR_Y, c = (lane % LPR_Y) * NBY;
            scratch_to_global<NBY>(scr + lane * NBY, a.dst + ((size_t)(uint32_t)(yw + r) * a.pitch_dst + (uint32_t)x0 * BPS + (uint32_t)c));
        }
        {
            const int r = lane / LPR_C, c = (lane % LPR_C) * NBC;
            scratch_to_global<NBC>(scr_c + lane * NBC, a.dst_uv + ((size_t)(uint32_t)((yw >> 1) + r) * a.pitch_dst_uv + (uint32_t)x0 * BPS + (uint32_t)c));
        }
    } else {
        // ragged tiles (last tile column / row) and unaligned destinations: sample by sample
#pragma unroll
        for (int j = 0; j < RW; j++) {
            const int y = yw + j;
            if (col_live && y < a.dh) {
                uint8_t *o = a.dst + ((size_t)(uint32_t)y * a.pitch_dst + (uint32_t)x * BPS);
                if constexpr (DEPTH == 10) {
                    const uint32_t wd = j & 1 ? outw[j / 2] >> 16 : outw[j / 2];
                    o[0] = (uint8_t)(wd & 255u), o[1] = (uint8_t)(wd >> 8);
                } else o[0] = (uint8_t)((uint32_t)out[j] >> 16);
            }
        }
        const int xc = x0 + (lane & ~1);  // the luma column this lane's chroma pairs belong to
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int y = yw + 2 * ((lane & 1) + 2 * s);  // the luma row
            if (xc < a.dw && y < a.dh && (lane & 1) + 2 * s < CR) {
                uint8_t *o = a.dst_uv + ((size_t)(uint32_t)(y >> 1) * a.pitch_dst_uv + (uint32_t)xc * BPS);
                if constexpr (DEPTH == 10) o[0] = (uint8_t)cw[s], o[1] = (uint8_t)(cw[s] >> 8), o[2] = (uint8_t)(cw[s] >> 16), o[3] = (uint8_t)(cw[s] >> 24);
                else o[0] = (uint8_t)((uint32_t)cu[s] >> 16), o[1] = (uint8_t)((uint32_t)cv[s] >> 16);
            }
        }
    }
    return true;
}

// Workgroups and tiles as k_warp_fused deals them (vstab_warp_fused.hip): block b -> XCD b % 8, every XCD one band of output rows,
// tall tiles first and half-height tiles for the last round of workgroup slots.
template <int RWB, int MODE, int DEPTH, int BLEND>
__global__ void __launch_bounds__(256, RWB == 8 ? 5 : 6) k_warp_planar(FusedArgs ta) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr int TH = 4 * RWB, TS = TH / 2;
    if (VSTAB_WARP_PRIO) __builtin_amdgcn_s_setprio(VSTAB_WARP_PRIO);
    const int k = (int)(blockIdx.x & 7u);
    const int y_lo = ta.band_y[k], y_sp = ta.split_y[k], y_hi = ta.band_y[k + 1];
    const int n_tall = ((y_sp - y_lo) / TH) * ta.tiles_x;
    const int idx = (int)(blockIdx.x >> 3);
    int x0, ys, n_half;
    if (idx < n_tall) {
        const int row = idx / ta.tiles_x;
        x0 = (idx - row * ta.tiles_x) * 64, ys = y_lo + row * TH;
        n_half = warp_tile_planar<RWB, RWB, MODE, true, DEPTH, BLEND>(ta, smem, x0, ys) ? 0 : 2;
    } else {
        const int i2 = idx - n_tall, row = i2 / ta.tiles_x;
        x0 = (i2 - row * ta.tiles_x) * 64, ys = y_sp + row * TS, n_half = 1;
        if (ys >= y_hi) return;  // uniform for the workgroup (before any barrier)
    }
#pragma unroll 1
    for (int i = 0; i < n_half; i++) warp_tile_planar<RWB, RWB / 2, MODE, false, DEPTH, BLEND>(ta, smem, x0, ys + i * TS);
}

// Launch; called by the C-ABI entry points (vstab_warp.hip, vstab_warp_p010.hip) after argument validation.  a.y / a.uv / a.dst / a.dst_uv:
// the four planes, pitches in bytes; depth 8 (NV12 bytes) or 10 (P010 words); src_vec_ok: source planes and pitches 16-byte aligned
// (else every pixel takes the global-memory path: correct, slow); dst_vec_ok: destination planes and pitches 16-byte aligned.
vstab_status launch_warp_planar(const WarpArgs &a, const float params[17], int map_mode, int depth, int blend, bool src_vec_ok, bool dst_vec_ok,
                                const float *rot_bottom, hipStream_t st) {
    FusedArgs ta;
    ta.w = a;
    ta.p32 = {params[0] * 32.0f, params[1] * 32.0f, params[2] * 32.0f, params[3] * 32.0f, params[10], params[13], params[16]};
    ta.src_vec_ok = src_vec_ok, ta.dst_vec_ok = dst_vec_ok;
    ta.qmap = nullptr, ta.qpitch = 0;
    for (int k = 0; k < 9; k++) ta.rs_d[k] = rot_bottom ? rot_bottom[k] - params[8 + k] : 0.0f;
    ta.rs_den = (float)(a.dh > 1 ? a.dh - 1 : 1);
    if (rot_bottom) map_mode = map_mode == VSTAB_MAP_CREATEMAP_CL_OPENCL ? (int)MAP_RS_CREATEMAP_CL_OPENCL : map_mode + (int)MAP_RS_CREATEMAP_CL;
#ifdef VSTAB_DEV
    ta.timing = nullptr, ta.lds_pad = 0;
    ta.ablate = getenv("VSTAB_ABLATE") ? atoi(getenv("VSTAB_ABLATE")) : 0;
#endif
    // 64 x 32 tiles; LDS per workgroup: header + scratch + 1.5 bytes (3 at 10 bits) per pixel of the box.  24 KB (8-bit) lets six
    // workgroups share a CU -- as many as the register budget admits -- and holds the largest boxes of a 4K fisheye frame.
    const int bps = depth == 10 ? 2 : 1;
    const long tiles32 = (long)div_up(a.dw, 64) * div_up(a.dh, 32);
    int rwb = tiles32 < 1536 ? 4 : 8;
#ifdef VSTAB_DEV
    if (const char *e = getenv("VSTAB_PLANAR_RWB")) rwb = atoi(e) == 4 ? 4 : 8;
#endif
    // (10 bits, 4K alone, binary16 blend: 48 KB -- the 8-bit kernel's box, three workgroups per CU -- 43.1 us; 40 KB, four: 37.0; 32 KB, five: 36.9)
    int lds_kb = rwb == 8 ? (bps == 2 ? 40 : 24) : 14 * bps;
    // workgroups a CU holds: by LDS, and by registers (the 64 x 32-tile kernels take up to 72: seven waves per SIMD; the others eight)
    const int resident = std::min({160 / lds_kb, rwb == 8 ? 7 : 8, 8});
    double tail_rounds = (double)div_up(a.dw, 64) * div_up(a.dh, 4 * rwb) > 256.0 * resident ? 0.25 : 0.0;
    // (measured at 4K: 0 / 0.25 / 0.5 / 1 rounds of half-height tiles at the end: 28.1 / 26.3 / 27.1 / 28.0 us; 20 to 28 KB of LDS: the same;
    //  1080p, everything resident at once: 64 x 16 tiles and no half-height tiles 11.3 us, 64 x 32 tiles 13.0: profiles/r05_planar_warp.txt)
#ifdef VSTAB_DEV
    if (const char *e = getenv("VSTAB_PLANAR_LDS_KB")) lds_kb = atoi(e);
    if (const char *e = getenv("VSTAB_PLANAR_TAIL")) tail_rounds = atof(e);
#endif
    const dim3 grid(tile_schedule(ta, rwb, lds_kb, tail_rounds));
    const size_t lds_bytes = (size_t)lds_kb * 1024;
    ta.lds_capacity_px = (int)((lds_bytes - 32 - 4 * 768 * (size_t)bps) * 2 / (3 * (size_t)bps));
    const LaunchEvents ev = take_launch_events();
#define VSTAB_LAUNCHP(R, M, D, B)                                                                                                 \
    do {                                                                                                                          \
        if (ev.start) hipExtLaunchKernelGGL((k_warp_planar<R, M, D, B>), grid, dim3(256), lds_bytes, st, ev.start, ev.stop, 0, ta); \
        else hipLaunchKernelGGL((k_warp_planar<R, M, D, B>), grid, dim3(256), lds_bytes, st, ta);                                  \
    } while (0)
#define VSTAB_LAUNCHP_M(M)                                                               \
    do {                                                                                 \
        if (depth == 10) {                                                               \
            if (blend == VSTAB_BLEND_FP16) {                                             \
                if (rwb == 8) VSTAB_LAUNCHP(8, M, 10, VSTAB_BLEND_FP16);                 \
                else VSTAB_LAUNCHP(4, M, 10, VSTAB_BLEND_FP16);                          \
            } else if (rwb == 8) VSTAB_LAUNCHP(8, M, 10, VSTAB_BLEND_EXACT);             \
            else VSTAB_LAUNCHP(4, M, 10, VSTAB_BLEND_EXACT);                             \
        } else if (rwb == 8) VSTAB_LAUNCHP(8, M, 8, 0);                                  \
        else VSTAB_LAUNCHP(4, M, 8, 0);                                                  \
    } while (0)
    switch (map_mode) {
        case VSTAB_MAP_CREATEMAP_CL: VSTAB_LAUNCHP_M(MAP_CREATEMAP_CL); break;
        case VSTAB_MAP_FISH_TO_RECT: VSTAB_LAUNCHP_M(MAP_FISH_TO_RECT); break;
        case VSTAB_MAP_FISH_TO_FISH: VSTAB_LAUNCHP_M(MAP_FISH_TO_FISH); break;
        case VSTAB_MAP_RECT_TO_RECT: VSTAB_LAUNCHP_M(MAP_RECT_TO_RECT); break;
        case VSTAB_MAP_RECT_TO_FISH: VSTAB_LAUNCHP_M(MAP_RECT_TO_FISH); break;
        case VSTAB_MAP_CREATEMAP_CL_OPENCL: VSTAB_LAUNCHP_M(MAP_CREATEMAP_CL_OPENCL); break;
        case MAP_RS_CREATEMAP_CL: VSTAB_LAUNCHP_M(MAP_RS_CREATEMAP_CL); break;
        case MAP_RS_CREATEMAP_CL_OPENCL: VSTAB_LAUNCHP_M(MAP_RS_CREATEMAP_CL_OPENCL); break;
        default: VSTAB_LAUNCHP_M(MAP_RS_FISH_TO_RECT); break;
    }
#undef VSTAB_LAUNCHP_M
#undef VSTAB_LAUNCHP
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}


// Kernels of this translation unit are one code object, loaded by the runtime at the first launch of any of them.  Touching one of them
// here (vstab_preload_kernels) moves that load to a moment the caller chooses.
vstab_status preload_planar_kernels() {
    hipFuncAttributes at;
    VSTAB_HIP_TRY(hipFuncGetAttributes(&at, reinterpret_cast<const void *>((&k_warp_planar<8, MAP_CREATEMAP_CL_OPENCL, 8, 0>))));
    return VSTAB_OK;
}

}  // namespace vstab
