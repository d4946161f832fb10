// vstab_track.hip -- camera-motion front-end for gfx950: image pyramid, Shi-Tomasi corner
// response + non-maximum suppression + compaction, pyramidal Lucas-Kanade tracker.
// Replaces the OpenCV calls at FrameSourceWarp.cpp:230 (goodFeaturesToTrack) and :252
// (calcOpticalFlowPyrLK).  Compiled with -ffp-contract=off: float results are bit-reproducible.
#include <algorithm>
#include <climits>

#include <hip/hip_ext.h>

#include "vstab_internal.hpp"
#include "vstab_track.hpp"

namespace vstab {

__device__ __forceinline__ int reflect101(int i, int n) {  // BORDER_REFLECT_101, any overshoot
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// One dword of a REFLECT_101-padded image row at columns gx .. gx+3 (gx a multiple of 4).
__device__ __forceinline__ uint32_t load4_reflect_row(const uint8_t *__restrict__ row, int w, int gx, bool vec_ok) {
    if (vec_ok && gx >= 0 && gx + 4 <= w) return *reinterpret_cast<const uint32_t *>(row + gx);
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) v |= (uint32_t)row[reflect101(gx + i, w)] << (8 * i);
    return v;
}

// One dword of a REFLECT_101-padded u8 image at (gx .. gx+3, gy), gx a multiple of 4.  Dwords that lie
// inside the row are one aligned load (also on border tiles); only dwords straddling the left /
// right image edge are assembled from bytes.
__device__ __forceinline__ uint32_t load4_reflect(const uint8_t *__restrict__ src, uint32_t pitch, int w, int h, int gx,
                                                  int gy, bool vec_ok) {
    const uint8_t *row = src + (uint32_t)reflect101(gy, h) * pitch;
    if (vec_ok && gx >= 0 && gx + 4 <= w) return *reinterpret_cast<const uint32_t *>(row + gx);
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) v |= (uint32_t)row[reflect101(gx + i, w)] << (8 * i);
    return v;
}

// =============================================================================================
// k_pyr_down -- cv::pyrDown as used by buildOpticalFlowPyramid: 5x5 binomial
// [1 4 6 4 1]^2, integer, (sum + 128) >> 8, REFLECT_101, dst = ((w+1)/2, (h+1)/2).
// Register-only: one thread produces 4 adjacent outputs of one row from five 16-byte row segments
// (aligned dword loads; the overlap between neighbouring threads is served by L1/L2).  The 25 taps of
// an output are accumulated with v_dot4_u32_u8 straight on the packed source dwords: the weight
// dword of row j holds k_j * (1 4 6 4 1) at the byte positions of the taps (<= 36, a byte), two
// dot products per output and row, so no byte is ever unpacked.  One dword store.  No LDS, no
// barriers: the kernel is a pure stream and overlaps with the LK kernel of the previous frame.
// =============================================================================================
__device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }

// Four adjacent outputs from the five 16-byte row segments that hold their taps (output c uses bytes 2c+2 .. 2c+6 of a segment), packed into
// one dword: the weight dword of row j holds k_j * (1 4 6 4 1) at the byte positions of the taps, two dot products per output and row.
__device__ __forceinline__ uint32_t pyr_down_dot4x4(const uint32_t (&d)[5][4]) {
    uint32_t acc[4] = {128u, 128u, 128u, 128u};
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const uint32_t k = j == 0 || j == 4 ? 1u : j == 2 ? 6u : 4u;
        // weight dwords (byte 0 = lowest address): taps 1 4 | 6 4 1 split over two dwords, or 1 4 6 4 | 1
        const uint32_t w_hi2 = (k << 16) | (4 * k << 24);          // (0, 0, k, 4k)
        const uint32_t w_lo3 = 6 * k | (4 * k << 8) | (k << 16);   // (6k, 4k, k, 0)
        const uint32_t w_all = k | (4 * k << 8) | (6 * k << 16) | (4 * k << 24);  // (k, 4k, 6k, 4k)
        acc[0] = udot4(d[j][1], w_lo3, udot4(d[j][0], w_hi2, acc[0]));
        acc[1] = udot4(d[j][2], k, udot4(d[j][1], w_all, acc[1]));
        acc[2] = udot4(d[j][2], w_lo3, udot4(d[j][1], w_hi2, acc[2]));
        acc[3] = udot4(d[j][3], k, udot4(d[j][2], w_all, acc[3]));
    }
    // result byte c = bits 8..15 of acc[c] (sum + 128 <= 255 * 256 + 128 < 2^16)
    const uint32_t p01 = __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0501u), p23 = __builtin_amdgcn_perm(acc[3], acc[2], 0x0c0c0501u);
    return __builtin_amdgcn_perm(p23, p01, 0x05040100u);
}

// outputs x0 .. x0+3 of row y, packed.  EDGE = false: the 16-byte window lies inside the row and everything is dword aligned.
// NEAR: sw >= 16 and sh >= 4, so that every tap is within one reflection of the image.
// COPY: the group also copies the source bytes it owns -- columns 2 x0 .. 2 x0 + 7 of rows 2 y and 2 y + 1, which it has loaded anyway
// (dwords 1 and 2 of window rows 2 and 3) -- to `cp` (pitch cpitch, 8-byte aligned rows when !EDGE): k_pack_pyr, the copy of an
// upstream frame into the ring and the first pyramid level in one pass over the luma plane.
template <bool EDGE, bool NEAR, bool COPY = false>
__device__ __forceinline__ uint32_t pyr_down_group4(const uint8_t *__restrict__ src, uint32_t spitch, int sw, int sh, int x0, int y, bool vec_ok,
                                                    uint8_t *__restrict__ cp = nullptr, uint32_t cpitch = 0) {
    const int sx0 = 2 * x0 - 4;  // the 16 bytes [sx0, sx0+16) hold the taps of outputs x0..x0+3: output c uses bytes 2c+2 .. 2c+6
    const uint8_t *row[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int ry = 2 * y - 2 + j;
        row[j] = src + (uint32_t)(NEAR ? min(abs(ry), 2 * sh - 2 - abs(ry)) : reflect101(ry, sh)) * spitch;
    }
    uint32_t d[5][4];
    if (!EDGE) {
        if (y >= 1 && 2 * y + 2 < sh) {
            // wave-uniform (a wave is one output row): the five source rows 2y - 2 .. 2y + 2 are inside the image, so their
            // addresses are one multiply and four pitch steps -- the reflection of every row (five quarter-rate 32-bit
            // multiplies, ten 64-bit adds, twenty min / max / sub) was a third of the kernel's vector instructions
            const uint8_t *r0 = src + (size_t)(uint32_t)(2 * y - 2) * spitch + sx0;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const uint4 v = *reinterpret_cast<const uint4 *>(r0 + (size_t)j * spitch);  // one 16-byte load (4-byte aligned: global loads take any alignment)
                d[j][0] = v.x, d[j][1] = v.y, d[j][2] = v.z, d[j][3] = v.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const uint32_t *p = reinterpret_cast<const uint32_t *>(row[j] + sx0);
                d[j][0] = p[0], d[j][1] = p[1], d[j][2] = p[2], d[j][3] = p[3];
            }
        }
    } else {
        // Whether a dword lies inside the row is the same for the five rows: one branch per dword column, the five (or
        // twenty byte) loads under it in flight together -- a branch per load would cost a memory latency per load.
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int gx = sx0 + 4 * q;
            if (vec_ok && gx >= 0 && gx + 4 <= sw) {
#pragma unroll
                for (int j = 0; j < 5; j++) d[j][q] = *reinterpret_cast<const uint32_t *>(row[j] + gx);
            } else {
                int col[4];
#pragma unroll
                for (int i = 0; i < 4; i++) col[i] = NEAR ? min(abs(gx + i), 2 * sw - 2 - abs(gx + i)) : reflect101(gx + i, sw);
#pragma unroll
                for (int j = 0; j < 5; j++)
                    d[j][q] = (uint32_t)row[j][col[0]] | ((uint32_t)row[j][col[1]] << 8) | ((uint32_t)row[j][col[2]] << 16) | ((uint32_t)row[j][col[3]] << 24);
            }
        }
    }
    if (COPY) {
#pragma unroll
        for (int jj = 2; jj <= 3; jj++) {
            const int r = 2 * y + jj - 2;
            if (r >= sh) break;  // (an odd height's last output row has no second source row)
            uint8_t *o = cp + (uint32_t)r * cpitch + (uint32_t)(2 * x0);
            if (!EDGE) {
                *reinterpret_cast<uint2 *>(o) = make_uint2(d[jj][1], d[jj][2]);
            } else {
#pragma unroll
                for (int b = 0; b < 8; b++)
                    if (2 * x0 + b < sw) o[b] = (uint8_t)((b < 4 ? d[jj][1] : d[jj][2]) >> (8 * (b & 3)));
            }
        }
    }
    return pyr_down_dot4x4(d);
}

template <bool EDGE, bool NEAR, bool COPY = false>
__device__ __forceinline__ void pyr_down_group(const uint8_t *__restrict__ src, uint32_t spitch, int sw, int sh, uint8_t *__restrict__ dst, size_t dpitch,
                                               int dw, int x0, int y, bool vec_ok, uint8_t *__restrict__ cp = nullptr, uint32_t cpitch = 0) {
    const uint32_t out = pyr_down_group4<EDGE, NEAR, COPY>(src, spitch, sw, sh, x0, y, vec_ok, cp, cpitch);
    uint8_t *o = dst + (size_t)((uint32_t)y * (uint32_t)dpitch) + x0;  // one 32-bit multiply (images are at most 32767 x 32767 bytes)
    if (!EDGE) {
        *reinterpret_cast<uint32_t *>(o) = out;
    } else {
        for (int c = 0; c < 4 && x0 + c < dw; c++) o[c] = (uint8_t)(out >> (8 * c));
    }
}

// Groups of 4 outputs [g_lo, g_hi) of every row are interior (64 groups x 4 rows per workgroup); the remaining groups --
// the first of a row, the last one to three, or all of them for an unaligned or tiny image -- are gathered in workgroups
// of their own, so that no wavefront of the bulk ever runs the byte path.  Those come first in the grid: they are the
// slow ones.
__global__ void __launch_bounds__(256) k_pyr_down(const uint8_t *__restrict__ src, size_t spitch, int sw, int sh,
                                                  uint8_t *__restrict__ dst, size_t dpitch, int dw, int dh, int vec_ok, int g_lo, int g_hi,
                                                  int nbx, int nb_edge, int n_groups, int near) {
    if ((int)blockIdx.x >= nb_edge) {
        const int b = blockIdx.x - nb_edge, by = b / nbx, bx = b - by * nbx;
        const int g = g_lo + bx * 64 + (threadIdx.x & 63), y = by * 4 + (threadIdx.x >> 6);
        if (g >= g_hi || y >= dh) return;
        pyr_down_group<false, true>(src, (uint32_t)spitch, sw, sh, dst, dpitch, dw, 4 * g, y, true);
    } else {
        const int n_edge = n_groups - (g_hi - g_lo);  // edge groups per row
        const int e = blockIdx.x * 256 + threadIdx.x;
        const int y = e / n_edge, i = e - y * n_edge;
        if (y >= dh) return;
        const int g = i < g_lo ? i : g_hi + (i - g_lo);
        if (near)
            pyr_down_group<true, true>(src, (uint32_t)spitch, sw, sh, dst, dpitch, dw, 4 * g, y, vec_ok != 0);
        else
            pyr_down_group<true, false>(src, (uint32_t)spitch, sw, sh, dst, dpitch, dw, 4 * g, y, vec_ok != 0);
    }
}

// k_pack_pyr -- the copy of an upstream NV12 frame into the library's ring AND the first pyramid level of its luma plane in one launch
// (vstab_frame.hold = 0: a decoder that recycles its surfaces): every group of k_pyr_down also stores the 8 x 2 source bytes it owns,
// and workgroups behind the pyramid's copy the chroma plane.  One pass over the luma instead of two, one kernel less on the read-ahead
// stream.  Same arithmetic, same bytes.
__global__ void __launch_bounds__(256) k_pack_pyr(const uint8_t *__restrict__ src, uint32_t spitch, int sw, int sh, uint8_t *__restrict__ dst, uint32_t dpitch,
                                                  int dw, int dh, int vec_ok, int g_lo, int g_hi, int nbx, int nb_edge, int n_groups, int near, int nb_pyr,
                                                  const uint8_t *__restrict__ uv, uint32_t uvpitch, uint8_t *__restrict__ ring, uint32_t rpitch, int uv_vec) {
    if ((int)blockIdx.x >= nb_pyr) {
        // chroma: sh / 2 rows of sw bytes behind the luma rows of the ring, 16 bytes per thread where everything is aligned
        uint8_t *cdst = ring + (uint32_t)sh * rpitch;
        const int rows = sh / 2;
        if (uv_vec) {
            const int vecs = sw / 16;
            for (int e = ((int)blockIdx.x - nb_pyr) * 256 + threadIdx.x; e < rows * vecs; e += ((int)gridDim.x - nb_pyr) * 256) {
                const int r = e / vecs, c = e - r * vecs;
                reinterpret_cast<uint4 *>(cdst + (uint32_t)r * rpitch)[c] = reinterpret_cast<const uint4 *>(uv + (uint32_t)r * uvpitch)[c];
            }
        } else {
            for (int e = ((int)blockIdx.x - nb_pyr) * 256 + threadIdx.x; e < rows * sw; e += ((int)gridDim.x - nb_pyr) * 256) {
                const int r = e / sw, c = e - r * sw;
                cdst[(uint32_t)r * rpitch + c] = uv[(uint32_t)r * uvpitch + c];
            }
        }
        return;
    }
    if ((int)blockIdx.x >= nb_edge) {
        const int b = blockIdx.x - nb_edge, by = b / nbx, bx = b - by * nbx;
        const int g = g_lo + bx * 64 + (threadIdx.x & 63), y = by * 4 + (threadIdx.x >> 6);
        if (g >= g_hi || y >= dh) return;
        pyr_down_group<false, true, true>(src, spitch, sw, sh, dst, dpitch, dw, 4 * g, y, true, ring, rpitch);
    } else {
        const int n_edge = n_groups - (g_hi - g_lo);  // edge groups per row
        const int e = blockIdx.x * 256 + threadIdx.x;
        const int y = e / n_edge, i = e - y * n_edge;
        if (y >= dh) return;
        const int g = i < g_lo ? i : g_hi + (i - g_lo);
        if (near)
            pyr_down_group<true, true, true>(src, spitch, sw, sh, dst, dpitch, dw, 4 * g, y, vec_ok != 0, ring, rpitch);
        else
            pyr_down_group<true, false, true>(src, spitch, sw, sh, dst, dpitch, dw, 4 * g, y, vec_ok != 0, ring, rpitch);
    }
}

// =============================================================================================
// k_pyr_down_x2 -- TWO pyramid levels in one launch (levels 2 and 3 of the LK pyramid from level 1): the small levels are
// launch- and latency-bound as kernels of their own (a 4K frame's level 3 is 480 x 270), and the prefetch stream paid three
// launches per frame.  A workgroup owns 16 x 10 outputs of the SECOND level: it computes the 40 x 23 first-level outputs around
// them straight from global memory with the arithmetic of k_pyr_down (four outputs per thread: 230 of the 256 threads, all
// loads of the workgroup in flight at once; into LDS, and to global memory for the 32 x 20 of them it owns), then the second
// level from those, four outputs per thread with the same dot products on the LDS dwords -- reflecting FIRST-LEVEL coordinates,
// as pyrDown of the stored first level does.  Integer sums: exact in any order.
// (The first version staged the 80 x 57 source bytes of a 14 x 12 tile in LDS, index arithmetic included, and produced one
// second-level output per thread: 0.58 M vector instructions per 4K frame and 6.3 us alone; this one 5.2 us (4.1 at 1080p).
// Tiles of 28 x 22 with three groups per thread need fewer instructions still but leave a 1080p frame 63 workgroups: 9 us.)
// =============================================================================================
constexpr int P2_TW = 16, P2_TH = 10;                   // second-level outputs per workgroup
constexpr int P2_MG = (2 * P2_TW + 5 + 3) / 4, P2_MW = 4 * P2_MG;  // first-level region: groups of 4 columns from 2 x0 - 4 on (taps of the tile: 2 x0 - 2 .. 2 x0 + 2 P2_TW)
constexpr int P2_MH = 2 * P2_TH + 3;                    // rows 2 y0 - 2 .. 2 y0 + 2 P2_TH
constexpr int P2_ITEMS = P2_MG * P2_MH, P2_ROUNDS = (P2_ITEMS + 255) / 256;
static_assert(P2_TW % 4 == 0 && 2 * P2_TW + 5 <= P2_MW, "the region holds the taps of the tile's last output");

template <bool NEAR>
__device__ __forceinline__ void pyr_down_x2_tile(const uint8_t *__restrict__ src, uint32_t spitch, int sw, int sh, uint8_t *__restrict__ mid, uint32_t mpitch, int mw,
                                                 int mh, uint8_t *__restrict__ dst, uint32_t dpitch, int dw, int dh, bool vec_ok, bool dst_vec_ok,
                                                 uint8_t (&smid)[P2_MH][P2_MW]) {
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * P2_TW, y0 = blockIdx.y * P2_TH;  // second-level origin of the tile
    const int mx0 = 2 * x0 - 4, my0 = 2 * y0 - 2;                 // first-level origin of the region
    // an interior workgroup: every first-level output of its region exists, and all their taps (source columns 2 mx0 - 4 .. 2 mx0 + 2 P2_MW + 3,
    // rows 2 my0 - 2 .. 2 my0 + 2 P2_MH) lie inside the source -- aligned dword loads, no reflection in either level, whole tile inside dst
    const bool interior = vec_ok && mx0 >= 2 && my0 >= 1 && mx0 + P2_MW <= mw && my0 + P2_MH <= mh && 2 * (mx0 + P2_MW) + 4 <= sw && 2 * (my0 + P2_MH) + 1 <= sh;  // uniform
    // ---- first level: P2_MG groups of 4 outputs x P2_MH rows, one group per thread (P2_ROUNDS = 1 with the 16 x 10 tile) --------------------------
    if (interior) {
        // every load of the thread's groups first (one memory latency however many rounds), then the arithmetic
        uint32_t d[P2_ROUNDS][5][4];
#pragma unroll
        for (int k = 0; k < P2_ROUNDS; k++) {
            const int it = min(tid + 256 * k, P2_ITEMS - 1);  // (a thread past the end loads the last group again and drops it)
            const int ry = it / P2_MG, g = it - ry * P2_MG;
            const uint8_t *r0 = src + (uint32_t)(2 * (my0 + ry) - 2) * spitch + (uint32_t)(2 * (mx0 + 4 * g) - 4);
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const uint4 v = *reinterpret_cast<const uint4 *>(r0 + (uint32_t)j * spitch);  // (16 bytes at a 4-byte aligned address: global loads take any alignment)
                d[k][j][0] = v.x, d[k][j][1] = v.y, d[k][j][2] = v.z, d[k][j][3] = v.w;
            }
        }
#pragma unroll
        for (int k = 0; k < P2_ROUNDS; k++) {
            const int it = tid + 256 * k;
            if (it >= P2_ITEMS) break;
            const int ry = it / P2_MG, g = it - ry * P2_MG;
            const uint32_t out = pyr_down_dot4x4(d[k]);
            *reinterpret_cast<uint32_t *>(&smid[ry][4 * g]) = out;
            // the tile owns first-level columns 2 x0 .. 2 x0 + 2 P2_TW - 1 (groups 1 .. P2_TW / 2) and rows 2 y0 .. 2 y0 + 2 P2_TH - 1
            if (g >= 1 && g <= P2_TW / 2 && ry >= 2 && ry < 2 + 2 * P2_TH) *reinterpret_cast<uint32_t *>(mid + (uint32_t)(my0 + ry) * mpitch + (mx0 + 4 * g)) = out;
        }
    } else {
#pragma unroll 1
        for (int k = 0; k < P2_ROUNDS; k++) {
            const int it = tid + 256 * k;
            if (it >= P2_ITEMS) break;
            const int ry = it / P2_MG, g = it - ry * P2_MG;
            const int my = my0 + ry, mx = mx0 + 4 * g;
            if (my < 0 || my >= mh || mx < 0 || mx >= mw) continue;  // (outside the first level: never read -- the second level reflects its coordinates into the image)
            const uint32_t out = pyr_down_group4<true, NEAR>(src, spitch, sw, sh, mx, my, vec_ok);
            *reinterpret_cast<uint32_t *>(&smid[ry][4 * g]) = out;
            if (g >= 1 && g <= P2_TW / 2 && ry >= 2 && ry < 2 + 2 * P2_TH) {
                uint8_t *o = mid + (uint32_t)my * mpitch + mx;
                if (vec_ok && mx + 4 <= mw) *reinterpret_cast<uint32_t *>(o) = out;
                else
                    for (int c = 0; c < 4 && mx + c < mw; c++) o[c] = (uint8_t)(out >> (8 * c));
            }
        }
    }
    __syncthreads();
    // ---- second level from the first-level region ----------------------------------------------------------------------------------------------
    if (interior) {
        // four outputs per item: their taps are bytes 8 g + 2 c + 2 .. + 6 of region rows 2 ty .. 2 ty + 4 -- the layout of pyr_down_dot4x4
        if (tid < (P2_TW / 4) * P2_TH) {
            const int ty = tid / (P2_TW / 4), g = tid - ty * (P2_TW / 4);
            uint32_t d[5][4];
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const uint2 lo = *reinterpret_cast<const uint2 *>(&smid[2 * ty + j][8 * g]), hi = *reinterpret_cast<const uint2 *>(&smid[2 * ty + j][8 * g + 8]);
                d[j][0] = lo.x, d[j][1] = lo.y, d[j][2] = hi.x, d[j][3] = hi.y;
            }
            const uint32_t out = pyr_down_dot4x4(d);
            uint8_t *o = dst + (uint32_t)(y0 + ty) * dpitch + (x0 + 4 * g);
            if (dst_vec_ok) *reinterpret_cast<uint32_t *>(o) = out;
            else
                for (int c = 0; c < 4; c++) o[c] = (uint8_t)(out >> (8 * c));
        }
    } else {
        // border workgroups: output by output, REFLECT_101 in first-level coordinates
        for (int e = tid; e < P2_TW * P2_TH; e += 256) {
            const int ty = e / P2_TW, tx = e - ty * P2_TW;
            const int x = x0 + tx, y = y0 + ty;
            if (x >= dw || y >= dh) continue;
            int col[5];
#pragma unroll
            for (int i = 0; i < 5; i++) col[i] = reflect101(2 * x - 2 + i, mw) - mx0;
            uint32_t acc = 128u;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const uint8_t *r = smid[reflect101(2 * y - 2 + j, mh) - my0];
                const uint32_t k = j == 0 || j == 4 ? 1u : j == 2 ? 6u : 4u;
                acc += k * ((uint32_t)r[col[0]] + 4u * r[col[1]] + 6u * r[col[2]] + 4u * r[col[3]] + r[col[4]]);
            }
            dst[(uint32_t)y * dpitch + x] = (uint8_t)(acc >> 8);
        }
    }
}

__global__ void __launch_bounds__(256) k_pyr_down_x2(const uint8_t *__restrict__ src, uint32_t spitch, int sw, int sh, uint8_t *__restrict__ mid,
                                                     uint32_t mpitch, int mw, int mh, uint8_t *__restrict__ dst, uint32_t dpitch, int dw, int dh, int vec_ok,
                                                     int dst_vec_ok, int near) {
    __shared__ __attribute__((aligned(16))) uint8_t smid[P2_MH][P2_MW];
    if (near) pyr_down_x2_tile<true>(src, spitch, sw, sh, mid, mpitch, mw, mh, dst, dpitch, dw, dh, vec_ok != 0, dst_vec_ok != 0, smid);
    else pyr_down_x2_tile<false>(src, spitch, sw, sh, mid, mpitch, mw, mh, dst, dpitch, dw, dh, vec_ok != 0, dst_vec_ok != 0, smid);
}

// =============================================================================================
// k_min_eig -- cornerMinEigenVal(blockSize 3, ksize 3): Sobel
// derivatives scaled by 1/(4*3*255) in the documented operation order, products, 3x3 box sum
// (exact in double), minimum eigenvalue in float; also reduces the frame maximum.
// 64 x 16 outputs per workgroup (4 per thread).  Source tile (halo 2, REFLECT_101) and derivative
// tile (halo 1) live in LDS.  The derivative the box filter needs at a position reflected across
// the image border is the derivative AT the mirrored position; computed from the mirrored tile it
// comes out with the sign of the mirrored axis flipped, so dx (dy) is negated for entries whose
// column (row) lies outside the image -- negation is exact, so the floats equal the direct form.
// =============================================================================================
constexpr int ME_TW = 64, ME_TH = 16, ME_SW = ME_TW + 8, ME_SH = ME_TH + 4;  // tile starts at ox - 4 (dword aligned)

__global__ void __launch_bounds__(256) k_min_eig(const uint8_t *__restrict__ src, size_t pitch, int w, int h,
                                                 float *__restrict__ eig, int *__restrict__ max_bits, int vec_ok) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[ME_SH][ME_SW];
    __shared__ float dxs[ME_TH + 2][ME_TW + 2 + 1], dys[ME_TH + 2][ME_TW + 2 + 1];
    __shared__ int bmax;
    const int tid = threadIdx.x;
    const int ox = blockIdx.x * ME_TW, oy = blockIdx.y * ME_TH;
    const int sx0 = ox - 4, sy0 = oy - 2;
    if (tid == 0) bmax = INT_MIN;
    for (int e = tid; e < ME_SH * (ME_SW / 4); e += 256) {
        const int ry = e / (ME_SW / 4), rd = e - ry * (ME_SW / 4);
        reinterpret_cast<uint32_t *>(&tile[ry][0])[rd] = load4_reflect(src, (uint32_t)pitch, w, h, sx0 + 4 * rd, sy0 + ry, vec_ok != 0);
    }
    __syncthreads();
    const float scale = (float)(1.0 / (4.0 * 3.0 * 255.0));
    const float k0 = 2.0f * scale, k1 = scale;
    // derivative entry (ry, rx) <-> image coordinate (oy - 1 + ry, ox - 1 + rx) <-> tile[ry + 1][rx + 3]
    for (int e = tid; e < (ME_TH + 2) * (ME_TW + 2); e += 256) {
        const int ry = e / (ME_TW + 2), rx = e - ry * (ME_TW + 2);
        const uint8_t *c = &tile[ry + 1][rx + 3];
        const int a00 = c[-ME_SW - 1], a01 = c[-ME_SW], a02 = c[-ME_SW + 1];
        const int a10 = c[-1], a12 = c[1];
        const int a20 = c[ME_SW - 1], a21 = c[ME_SW], a22 = c[ME_SW + 1];
        const float d0 = (float)(a02 - a00), d1 = (float)(a12 - a10), d2 = (float)(a22 - a20);
        float dx = (d0 + d2) * k1 + d1 * k0;
        const float s0 = (float)a01 * k0 + ((float)a00 + (float)a02) * k1;
        const float s2 = (float)a21 * k0 + ((float)a20 + (float)a22) * k1;
        float dy = s2 - s0;
        const int gx = ox - 1 + rx, gy = oy - 1 + ry;
        if (gx < 0 || gx >= w) dx = -dx;
        if (gy < 0 || gy >= h) dy = -dy;
        dxs[ry][rx] = dx, dys[ry][rx] = dy;
    }
    __syncthreads();
    const int q = tid & 15, ty = tid >> 4;
    const int x = ox + 4 * q, y = oy + ty;
    int best = INT_MIN;
    if (x < w && y < h) {
        float a_[3][6], b_[3][6];
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int i = 0; i < 6; i++) a_[j][i] = dxs[ty + j][4 * q + i], b_[j][i] = dys[ty + j][4 * q + i];
        float ev[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            double sxx = 0, sxy = 0, syy = 0;
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const float a = a_[j][c + i], b = b_[j][c + i];
                    sxx += (double)(a * a), sxy += (double)(a * b), syy += (double)(b * b);
                }
            const float a = (float)sxx * 0.5f, b = (float)sxy, cc = (float)syy * 0.5f;
            ev[c] = (a + cc) - sqrtf((a - cc) * (a - cc) + b * b);
            if (x + c < w) best = max(best, __float_as_int(ev[c]));
        }
        float *o = eig + (size_t)y * w + x;
        if (vec_ok && (w & 3) == 0 && x + 4 <= w) {
            *reinterpret_cast<float4 *>(o) = make_float4(ev[0], ev[1], ev[2], ev[3]);
        } else {
            for (int c = 0; c < 4 && x + c < w; c++) o[c] = ev[c];
        }
    }
    // frame maximum: DPP max inside each 16-lane row, one LDS atomic per row, one global atomic per block
    best = max(best, __builtin_amdgcn_update_dpp(INT_MIN, best, 0x111, 0xf, 0xf, false));
    best = max(best, __builtin_amdgcn_update_dpp(INT_MIN, best, 0x112, 0xf, 0xf, false));
    best = max(best, __builtin_amdgcn_update_dpp(INT_MIN, best, 0x114, 0xf, 0xf, false));
    best = max(best, __builtin_amdgcn_update_dpp(INT_MIN, best, 0x118, 0xf, 0xf, false));
    if (q == 15) atomicMax(&bmax, best);
    __syncthreads();
    if (tid == 0) atomicMax(max_bits, bmax);
}

// =============================================================================================
// k_corner_candidates -- goodFeaturesToTrack steps 4-5: threshold at
// quality*max (THRESH_TOZERO, strict >), 3x3 dilate-compare, interior pixels only.  A candidate is
// emitted as the 64-bit key (float bits << 32 | raster index): sorting keys descending gives
// OpenCV's order (value descending, ties -> later raster position first).
// =============================================================================================
__global__ void __launch_bounds__(256) k_corner_candidates(const float *__restrict__ eig, int w, int h,
                                                           const int *__restrict__ max_bits, double quality,
                                                           unsigned long long *__restrict__ keys,
                                                           unsigned int *__restrict__ count, unsigned int cap) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x < 1 || y < 1 || x >= w - 1 || y >= h - 1) return;
    const float thr = (float)((double)__int_as_float(*max_bits) * quality);
    const float *p = eig + (size_t)y * w + x;
    const float v = p[0];
    if (!(v > thr)) return;
    float m = v;
    m = fmaxf(m, p[-w - 1]), m = fmaxf(m, p[-w]), m = fmaxf(m, p[-w + 1]);
    m = fmaxf(m, p[-1]), m = fmaxf(m, p[1]);
    m = fmaxf(m, p[w - 1]), m = fmaxf(m, p[w]), m = fmaxf(m, p[w + 1]);
    if (v != m) return;
    const unsigned int slot = atomicAdd(count, 1u);
    if (slot < cap) keys[slot] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned int)(y * w + x);
}

// =============================================================================================
// k_corners_fused -- cornerMinEigenVal + the threshold / 3x3 non-maximum test of goodFeaturesToTrack in ONE pass
// over the image: the eigenvalue map never goes to HBM.  A workgroup owns 64 x 31 output
// pixels and evaluates the eigenvalue on the 66 x 33 pixels around them (the halo ring is recomputed, 1.11x), so the
// 3x3 maximum test needs nothing from a neighbour.
//   load    72 x 38 source bytes (REFLECT_101) -> LDS
//   deriv   68 x 36 Sobel pairs: a thread owns 4 columns x 3 rows and shares the per-row terms D = I(x+1) - I(x-1),
//           S = I(x) k0 + (I(x-1) + I(x+1)) k1 between them (5 source rows for 3 output rows); same float operation
//           order as k_min_eig -> identical bits; floats -> LDS
//   eig     a thread owns 3 x 3 pixels: the 5 x 5 products go to double once, the 3-row column sums are shared by
//           the three pixels of a row (box sums of float products are exact in double in any order: <= 48
//           significant bits), 33 instead of 81 double additions per pixel; eigenvalue in float as k_min_eig
//   nms     eigenvalues -> LDS; 3x3 maximum with v_max3, threshold, wave-aggregated append of 64-bit keys
// The threshold quality * max(frame) is not known until every tile is done, so a tile filters with a LOWER bound of
// it -- quality * max(own tile, frame maximum published so far) -- and k_filter_keys applies the final threshold to
// the survivors.  Which candidates survive the first filter depends on timing; the set that survives the second
// does not (lower bound <= final threshold, monotone rounding), and the host sorts the keys.
// =============================================================================================
constexpr int CF_TW = 64, CF_TH = 31;                  // output pixels per tile
constexpr int CF_EH = CF_TH + 2;                      // eigenvalue region: 66 x 33, image (oy - 1 .., ox - 1 ..)
constexpr int CF_DH = 36, CF_DP = 68;                 // derivative region: 68 x 36, image (oy - 2 .., ox - 2 ..); 35 rows used
constexpr int CF_SW = 72, CF_SH = 38;                  // source tile: image (oy - 3 .., ox - 4 ..)
constexpr int CF_EP = 67;
constexpr int CF_SLOTS = 256;                          // key slots per tile (a tile holds at most 1984 / 4 strict 3x3 maxima; fine noise reaches ~220)

__device__ __forceinline__ float ubyte_f32(uint32_t v, int b) { return (float)((v >> (8 * b)) & 255u); }  // v_cvt_f32_ubyteN

// threshold + 3x3 maximum test over the eigenvalues in LDS.  DENSE = false: survivors are appended to the tile's slots
// (*bcount counts all of them, also those past the last slot); DENSE = true: every pixel of the tile is written, the
// eigenvalue for a survivor and -inf otherwise.
template <bool DENSE>
__device__ __forceinline__ void cf_nonmax(const float (&es)[CF_EH][CF_EP], int tid, int ox, int oy, int w, int h, float thr_lb,
                                          unsigned long long *__restrict__ my_slots, unsigned int *bcount, float *__restrict__ dense) {
    const int tx = tid & 63, r0 = (tid >> 6) * 8;
    const int x = ox + tx;
    float rm0 = fmaxf(fmaxf(es[r0][tx], es[r0][tx + 1]), es[r0][tx + 2]);
    float c1 = es[r0 + 1][tx + 1];
    float rm1 = fmaxf(fmaxf(es[r0 + 1][tx], c1), es[r0 + 1][tx + 2]);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int ty = r0 + r;
        if (ty >= CF_TH) break;  // wave-uniform (the last wave owns 7 rows)
        const float c2 = es[ty + 2][tx + 1];
        const float rm2 = fmaxf(fmaxf(es[ty + 2][tx], c2), es[ty + 2][tx + 2]);
        const float m = fmaxf(fmaxf(rm0, rm1), rm2);
        const int y = oy + ty;
        const bool cand = x >= 1 && y >= 1 && x < w - 1 && y < h - 1 && c1 > thr_lb && c1 == m;
        if (DENSE) {
            dense[ty * CF_TW + tx] = cand ? c1 : -__builtin_inff();
        } else {
            const unsigned long long ballot = __ballot(cand);
            if (ballot) {
                unsigned int base = 0;
                if (tx == 0) base = atomicAdd(bcount, (unsigned int)__popcll(ballot));  // LDS
                base = __builtin_amdgcn_readfirstlane(base);
                const unsigned int slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0));
                if (cand && slot < CF_SLOTS) my_slots[slot] = ((unsigned long long)__float_as_uint(c1) << 32) | (unsigned int)(y * w + x);
            }
        }
        rm0 = rm1, rm1 = rm2, c1 = c2;
    }
}

__global__ void __launch_bounds__(256) k_corners_fused(const uint8_t *__restrict__ src, size_t pitch, int w, int h, double quality,
                                                       unsigned int *__restrict__ max_key, unsigned long long *__restrict__ slots,
                                                       unsigned int *__restrict__ tile_counts, float *__restrict__ spill, int vec_ok) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[CF_SH][CF_SW];
    __shared__ __attribute__((aligned(16))) float dxs[CF_DH][CF_DP], dys[CF_DH][CF_DP];
    __shared__ float es[CF_EH][CF_EP];
    __shared__ int bmax;
    __shared__ unsigned int bcount;
    const int tid = threadIdx.x;
    const int ox = blockIdx.x * CF_TW, oy = blockIdx.y * CF_TH;
    // frame maximum published so far (biased bits): a lower bound of the final one.  Device-scope load: the atomics
    // of other XCDs do not pass through this XCD's L2.
    const unsigned int seen = __hip_atomic_load(max_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) bmax = INT_MIN, bcount = 0;
    for (int e = tid; e < CF_SH * (CF_SW / 4); e += 256) {
        const int ry = e / (CF_SW / 4), rd = e - ry * (CF_SW / 4);
        reinterpret_cast<uint32_t *>(&tile[ry][0])[rd] = load4_reflect(src, (uint32_t)pitch, w, h, ox - 4 + 4 * rd, oy - 3 + ry, vec_ok != 0);
    }
    __syncthreads();
    const float scale = (float)(1.0 / (4.0 * 3.0 * 255.0));
    const float k0 = 2.0f * scale, k1 = scale;
    if (tid < 17 * 12) {
        const int g = tid / 17, c = tid - g * 17;
        // derivative entry (r, q) <-> image (oy - 2 + r, ox - 2 + q) <-> tile[r + 1][q + 2]; this thread: q = 4c .. 4c+3, r = 3g .. 3g+2
        float D[5][4], S[5][4];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const uint32_t *row = reinterpret_cast<const uint32_t *>(&tile[3 * g + j][4 * c]);
            const uint32_t lo = row[0], hi = row[1];  // bytes 4c .. 4c+7; columns q-1 .. q+4 are bytes 1 .. 6
            const float f[6] = {ubyte_f32(lo, 1), ubyte_f32(lo, 2), ubyte_f32(lo, 3), ubyte_f32(hi, 0), ubyte_f32(hi, 1), ubyte_f32(hi, 2)};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                D[j][i] = f[i + 2] - f[i];
                S[j][i] = f[i + 1] * k0 + (f[i] + f[i + 2]) * k1;
            }
        }
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int gy = oy - 2 + 3 * g + r;
            const bool fy = gy < 0 || gy >= h;
            float dx[4], dy[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int gx = ox - 2 + 4 * c + i;
                dx[i] = (D[r][i] + D[r + 2][i]) * k1 + D[r + 1][i] * k0;
                dy[i] = S[r + 2][i] - S[r][i];
                if (gx < 0 || gx >= w) dx[i] = -dx[i];
                if (fy) dy[i] = -dy[i];
            }
            *reinterpret_cast<float4 *>(&dxs[3 * g + r][4 * c]) = make_float4(dx[0], dx[1], dx[2], dx[3]);
            *reinterpret_cast<float4 *>(&dys[3 * g + r][4 * c]) = make_float4(dy[0], dy[1], dy[2], dy[3]);
        }
    }
    __syncthreads();
    int best = INT_MIN;
    if (tid < 22 * 11) {
        const int by = tid / 22, bx = tid - by * 22;
        // eigenvalue entry (ey, ex) <-> image (oy - 1 + ey, ox - 1 + ex) <-> derivative rows ey .. ey+2, columns ex .. ex+2
        float a_[5][5], b_[5][5];
#pragma unroll
        for (int j = 0; j < 5; j++)
#pragma unroll
            for (int i = 0; i < 5; i++) a_[j][i] = dxs[3 * by + j][3 * bx + i], b_[j][i] = dys[3 * by + j][3 * bx + i];
        float sum[3][3][3];  // [quantity][row][column] box sums rounded to float
#pragma unroll
        for (int q = 0; q < 3; q++) {
            double col[3][5];
#pragma unroll
            for (int j = 0; j < 5; j++)
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    const float u = q == 2 ? b_[j][i] : a_[j][i], v = q == 0 ? a_[j][i] : b_[j][i];
                    const double p = (double)(u * v);
#pragma unroll
                    for (int e = 0; e < 3; e++)
                        if (j >= e && j <= e + 2) col[e][i] = j == e ? p : col[e][i] + p;
                }
#pragma unroll
            for (int e = 0; e < 3; e++)
#pragma unroll
                for (int i = 0; i < 3; i++) sum[q][e][i] = (float)((col[e][i] + col[e][i + 1]) + col[e][i + 2]);
        }
#pragma unroll
        for (int e = 0; e < 3; e++)
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const float a = sum[0][e][i] * 0.5f, b = sum[1][e][i], cc = sum[2][e][i] * 0.5f;
                const float ev = (a + cc) - sqrtf((a - cc) * (a - cc) + b * b);
                es[3 * by + e][3 * bx + i] = ev;
                const int gx = ox - 1 + 3 * bx + i, gy = oy - 1 + 3 * by + e;
                if (gx >= 0 && gx < w && gy >= 0 && gy < h) best = max(best, __float_as_int(ev));
            }
    }
    // tile maximum (same int-bit order as k_min_eig)
    best = max(best, __builtin_amdgcn_update_dpp(INT_MIN, best, 0x111, 0xf, 0xf, false));
    best = max(best, __builtin_amdgcn_update_dpp(INT_MIN, best, 0x112, 0xf, 0xf, false));
    best = max(best, __builtin_amdgcn_update_dpp(INT_MIN, best, 0x114, 0xf, 0xf, false));
    best = max(best, __builtin_amdgcn_update_dpp(INT_MIN, best, 0x118, 0xf, 0xf, false));
    best = max(best, __builtin_amdgcn_update_dpp(INT_MIN, best, 0x142, 0xa, 0xf, false));
    best = max(best, __builtin_amdgcn_update_dpp(INT_MIN, best, 0x143, 0xc, 0xf, false));
    if ((tid & 63) == 63) atomicMax(&bmax, best);
    __syncthreads();
    const int tile_max = bmax;
    // Every workgroup hitting one address costs ~11 ns apiece on this part (all XCDs meet at the memory side):
    // only a tile that raises the maximum it saw publishes.
    if (tid == 0 && ((unsigned int)tile_max ^ 0x80000000u) > seen) atomicMax(max_key, (unsigned int)tile_max ^ 0x80000000u);
    // lower bound of the frame threshold; only meaningful for a non-negative maximum (float order == int order)
    const int lb_bits = max(tile_max, (int)(seen ^ 0x80000000u));
    const float thr_lb = lb_bits >= 0 ? (float)((double)__int_as_float(lb_bits) * quality) : -__builtin_inff();
    // 3x3 maximum test: lane = column, a wave walks 8 rows with a rolling row-maximum
    const int tile_idx = blockIdx.y * gridDim.x + blockIdx.x;
    cf_nonmax<false>(es, tid, ox, oy, w, h, thr_lb, slots + (size_t)tile_idx * CF_SLOTS, &bcount, nullptr);
    __syncthreads();
    const unsigned int n = bcount;
    if (tid == 0) tile_counts[tile_idx] = n;
    // A tile with more survivors than slots (an eigenvalue plateau: a smooth ramp, a periodic texture) leaves them as a
    // dense 64 x 31 map instead (-inf = not a survivor); k_filter_keys scans that with the final threshold.
    if (n > CF_SLOTS) cf_nonmax<true>(es, tid, ox, oy, w, h, thr_lb, nullptr, nullptr, spill + (size_t)tile_idx * (CF_TW * CF_TH));
}

// k_filter_keys -- the final threshold quality * max(frame) over the survivors of k_corners_fused.  A workgroup
// gathers the keys of FK_TILES tiles in LDS and appends them with ONE global atomic; a tile that spilled (dense map) is
// scanned row by row and appended per wavefront.  counts[0] = keys kept (may exceed cap_out: the caller then re-runs
// with the two-pass detector, whose key buffer grows), counts[1] = number of spilled tiles (statistics).
constexpr int FK_TILES = 16;
__global__ void __launch_bounds__(256) k_filter_keys(const unsigned long long *__restrict__ slots, const unsigned int *__restrict__ tile_counts,
                                                     const float *__restrict__ spill, int n_tiles, int tiles_x, int w,
                                                     const unsigned int *__restrict__ max_key, double quality,
                                                     unsigned long long *__restrict__ out, unsigned int *__restrict__ counts, unsigned int cap_out) {
    __shared__ unsigned long long kept[FK_TILES * CF_SLOTS];
    __shared__ unsigned int n_kept, base, n_spilled;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) n_kept = 0, n_spilled = 0;
    __syncthreads();
    const float thr = (float)((double)__int_as_float((int)(*max_key ^ 0x80000000u)) * quality);
    for (int k = 0; k < FK_TILES / 4; k++) {
        const int t = blockIdx.x * FK_TILES + wave * (FK_TILES / 4) + k;
        if (t >= n_tiles) break;
        const unsigned int n = tile_counts[t];
        if (n > CF_SLOTS) {
            if (lane == 0) atomicAdd(&n_spilled, 1u);
            const int ty0 = t / tiles_x, ox = (t - ty0 * tiles_x) * CF_TW, oy = ty0 * CF_TH;
            float v[CF_TH];  // all 31 rows in flight at once: one memory latency, not 31
#pragma unroll
            for (int r = 0; r < CF_TH; r++) v[r] = spill[(size_t)t * (CF_TW * CF_TH) + r * CF_TW + lane];
#pragma unroll
            for (int r = 0; r < CF_TH; r++) {
                const bool keep = v[r] > thr;
                const unsigned long long ballot = __ballot(keep);
                if (ballot) {
                    unsigned int b = 0;
                    if (lane == 0) b = atomicAdd(&counts[0], (unsigned int)__popcll(ballot));
                    b = __builtin_amdgcn_readfirstlane(b);
                    const unsigned int slot = b + __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0));
                    if (keep && slot < cap_out) out[slot] = ((unsigned long long)__float_as_uint(v[r]) << 32) | (unsigned int)((oy + r) * w + ox + lane);
                }
            }
            continue;
        }
        for (unsigned int i = lane; i < ((n + 63) & ~63u); i += 64) {
            unsigned long long key = 0;
            bool keep = false;
            if (i < n) {
                key = slots[(size_t)t * CF_SLOTS + i];
                keep = __uint_as_float((unsigned int)(key >> 32)) > thr;
            }
            const unsigned long long ballot = __ballot(keep);
            if (ballot) {
                unsigned int b = 0;
                if (lane == 0) b = atomicAdd(&n_kept, (unsigned int)__popcll(ballot));
                b = __builtin_amdgcn_readfirstlane(b);
                if (keep) kept[b + __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0))] = key;
            }
        }
    }
    __syncthreads();
    const unsigned int n = n_kept;
    if (tid == 0) {
        if (n) base = atomicAdd(&counts[0], n);
        if (n_spilled) atomicAdd(&counts[1], n_spilled);
    }
    __syncthreads();
    for (unsigned int i = tid; i < n; i += 256)
        if (base + i < cap_out) out[base + i] = kept[i];
}

// =============================================================================================
// k_lk_track -- LKTrackerInvoker for all pyramid levels, one wavefront per
// feature.  Window 21x21 = 441 pixels -> 7 per lane.  Per level: the 24x24 neighbourhood of the
// previous image goes to LDS (REFLECT_101 padding), Scharr derivatives are computed on the fly
// for the 22x22 taps (zero outside the image, as the reference's zero-padded derivative buffer),
// the patch (I, Ix, Iy as int16 x32 fixed point) lives in registers, and the Gauss-Newton loop
// stages a 22x22 block of the next image per iteration.  All sums are exact integers reduced
// with wave shuffles (order free), converted once to float; the 2x2 solve follows the reference's
// float operation order.
// =============================================================================================
constexpr int LKW = 21, LKR = 24, LKT = 22;
constexpr int LKJM = 5, LKJR = LKT + 2 * LKJM;  // next-image region: 22x22 taps + 5 px of slack each side

// Full-wave integer sum by DPP (row_shr 1,2,4,8 then row_bcast 15/31); the total is read from
// lane 63 and broadcast through an SGPR.  Integer addition is associative, so the order is free.
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return __builtin_amdgcn_readlane(v, 63);
}
#define LK_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))
// 2^-l as a float, exactly what (float)(1.0 / (double)(1 << l)) gives -- without a double-precision division in the kernel
__device__ __forceinline__ float lk_level_scale(int l) { return __uint_as_float((uint32_t)(127 - l) << 23); }
// 32-bit integer multiplies run at a quarter of the rate of the 24-bit ones, and every product here has operands well inside 24 bits
// (pixels < 2^8, weights <= 2^14, Scharr derivatives and interpolated patch values < 2^15) and a result inside 32
__device__ __forceinline__ int lk_mul(int a, int b) { return __mul24(a, b); }
// a * b + c with a 24-bit multiply, spelled out: where b is a constant the compiler turns __mul24 back into a full 32-bit multiply
// (it cannot see that the LDS values are small) and the Scharr taps became v_mul_lo_u32 / v_mad_u64_u32 pairs
__device__ __forceinline__ int lk_mad(int a, int b, int c) {
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// e / 22 and k / 21 for the indices of the 22 x 22 tap block and the 21 x 21 window (< 600): one 24-bit multiply and a shift
// instead of the 64-bit multiply-high the compiler emits for a division by a constant
constexpr bool lk_div_magic_ok(int d, int magic) {
    for (int e = 0; e < 600; e++)
        if (((e * magic) >> 16) != e / d) return false;
    return true;
}
static_assert(lk_div_magic_ok(22, 2979) && lk_div_magic_ok(21, 3121), "division constants");
__device__ __forceinline__ int lk_div22(int e) { return __mul24(e, 2979) >> 16; }
__device__ __forceinline__ int lk_div21(int k) { return __mul24(k, 3121) >> 16; }

// k_lk_track: ONE WORKGROUP OF 4 WAVES PER FEATURE.  The tracker is latency bound (<= 200 features, a dependent
// Gauss-Newton chain of up to 4 x 30 iterations per frame pair): a wave alone on its SIMD issues a dependent instruction
// only every ~8 cycles, so what counts is the NUMBER of instructions on the chain, not the arithmetic in them.  Four waves
// stage the image blocks (one memory latency for all of them) and prepare the previous image's side of the four pyramid
// levels, one level per wave; the iterations of a level run on one wave with seven window pixels per lane (see below).
// Integer sums are exact whatever the decomposition: per-lane partials are split into a signed high part and a 16-bit low
// part so that every wave sum stays inside int32; hi * 65536 + lo is exact in double and the one double -> float
// conversion equals (float)(int64 total).
constexpr int LK_THREADS = 256, LK_WAVES = 4;
#ifndef VSTAB_LK_PRIO
#define VSTAB_LK_PRIO 1
#endif
constexpr int LK_WPAD = (LKW * LKW + 63) & ~63;  // window pixels rounded up to whole waves

// The 2 x 2 matrix of a level from the exact sums of Ix Ix, Ix Iy, Iy Iy (as floats): A11, A12, A22, 1 / D and whether the level is
// rejected (minEig < 1e-4 or D < FLT_EPSILON), in the reference's float operation order.  Evaluated by whoever prepared the level --
// for the lower levels a wave in the shadow of the top level's iterations -- so that the square root and the two divisions are not
// on the iterating wave's dependent chain; the same instructions give the same bits wherever they run.
__device__ __forceinline__ void lk_level_matrix(float s0, float s1, float s2, float *out) {
    const float FLT_SCALE = 1.0f / (1 << 20);
    const float A11 = s0 * FLT_SCALE, A12 = s1 * FLT_SCALE, A22 = s2 * FLT_SCALE;
    const float D = A11 * A22 - A12 * A12;
    const float minEig = ((A22 + A11) - sqrtf((A11 - A22) * (A11 - A22) + (4.f * A12) * A12)) / (float)(2 * LKW * LKW);
    out[0] = A11, out[1] = A12, out[2] = A22, out[3] = 1.f / D;
    out[4] = minEig < 1e-4f || D < 1.1920928955078125e-7f ? 1.0f : 0.0f;
}

// The 2 * NQ wave reductions advance in lockstep: every DPP step is applied to all of them before the next one, so the
// two wait states a DPP read needs behind the write of its source are filled by the other chains instead of s_nop
// (written one chain after the other, the compiler serialised the first pair: 12 steps with a nop each).
template <int N>
__device__ __forceinline__ void wave_sums_i32(int (&t)[N]) {
#define VSTAB_DPP_STEP(ctrl, rows)                                                          \
    _Pragma("unroll") for (int i = 0; i < N; i++) t[i] += __builtin_amdgcn_update_dpp(0, t[i], ctrl, rows, 0xf, false)
    VSTAB_DPP_STEP(0x111, 0xf);
    VSTAB_DPP_STEP(0x112, 0xf);
    VSTAB_DPP_STEP(0x114, 0xf);
    VSTAB_DPP_STEP(0x118, 0xf);
    VSTAB_DPP_STEP(0x142, 0xa);
    VSTAB_DPP_STEP(0x143, 0xc);
#undef VSTAB_DPP_STEP
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = __builtin_amdgcn_readlane(t[i], 63);
}

// Staging in two phases -- every global load of a block is issued before the first LDS store waits for one -- so
// a block costs ONE memory latency instead of one per loop trip.  The tracker is a dependent chain at one
// workgroup per CU: exposed latency is what it is made of (measured: staging was half of a workgroup's time).
template <int SIDE, int THREADS = LK_THREADS>
struct LkStage {
    // SIDE x SIDE bytes as SIDE * SIDE / 4 dwords: a thread fetches FOUR consecutive pixels of a row with one (unaligned)
    // dword load and writes them to LDS as four ints with one 16-byte store -- a quarter of the loads, index arithmetic and LDS
    // stores of the byte-by-byte form it replaces (which is kept for blocks that touch the image border: REFLECT_101 per byte).
    static_assert(SIDE % 4 == 0, "rows are split into dwords");
    static constexpr int ROWW = SIDE / 4, NW = SIDE * ROWW, N = (NW + THREADS - 1) / THREADS;
    uint32_t v[N];
    __device__ __forceinline__ void load(const uint8_t *img, uint32_t pitch, int w, int h, int x0, int y0, int tid) {
        const bool interior = x0 >= 0 && y0 >= 0 && x0 + SIDE <= w && y0 + SIDE <= h;  // uniform
#pragma unroll
        for (int k = 0; k < N; k++) {
            const int e = tid + THREADS * k;
            v[k] = 0;
            if (e < NW) {
                const int ry = e / ROWW, rx = 4 * (e - ry * ROWW);
                if (interior) {
                    // (row * pitch with the 24-bit multiplier: rows < 2^15, pitches < 2^24 -- launch_lk checks -- and planes < 4 GiB)
                    __builtin_memcpy(&v[k], img + __umul24((uint32_t)(y0 + ry), pitch) + (uint32_t)(x0 + rx), 4);  // one global_load_dword (any alignment)
                } else {
                    const uint8_t *row = img + __umul24((uint32_t)reflect101(y0 + ry, h), pitch);
                    v[k] = (uint32_t)row[reflect101(x0 + rx, w)] | ((uint32_t)row[reflect101(x0 + rx + 1, w)] << 8) |
                           ((uint32_t)row[reflect101(x0 + rx + 2, w)] << 16) | ((uint32_t)row[reflect101(x0 + rx + 3, w)] << 24);
                }
            }
        }
    }
    // a block that lies inside the image (the caller checked): no border handling, a handful of instructions per dword
    __device__ __forceinline__ void load_interior(const uint8_t *img, uint32_t pitch, int x0, int y0, int tid) {
#pragma unroll
        for (int k = 0; k < N; k++) {
            const int e = min(tid + THREADS * k, NW - 1);  // (a thread past the end loads the last dword again and never stores it)
            const int ry = e / ROWW, rx = 4 * (e - ry * ROWW);
            __builtin_memcpy(&v[k], img + __umul24((uint32_t)(y0 + ry), pitch) + (uint32_t)(x0 + rx), 4);
        }
    }
    // the OUT x OUT window of the block whose corner sits at (dx, dy) of it, as a row-major OUT x OUT int array (the window's place in the
    // block is only known after the block was fetched: the previous-image neighbourhoods fetched ahead, k_lk_track)
    template <int OUT>
    __device__ __forceinline__ void store_window(int *dst, int tid, int dx, int dy) const {
#pragma unroll
        for (int k = 0; k < N; k++) {
            const int e = tid + THREADS * k;
            const int ry = e / ROWW, rx = 4 * (e - ry * ROWW);
            const int Y = ry - dy, X = rx - dx;
            if (e < NW && (unsigned)Y < (unsigned)OUT) {
                int *o = dst + lk_mul(Y, OUT) + X;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if ((unsigned)(X + i) < (unsigned)OUT) o[i] = (int)((v[k] >> (8 * i)) & 255u);
            }
        }
    }
    __device__ __forceinline__ void store(int *dst, int tid) const {  // dst 16-byte aligned
#pragma unroll
        for (int k = 0; k < N; k++) {
            const int e = tid + THREADS * k;
            if (e < NW) *reinterpret_cast<int4 *>(dst + 4 * e) = make_int4((int)(v[k] & 255u), (int)((v[k] >> 8) & 255u), (int)((v[k] >> 16) & 255u), (int)(v[k] >> 24));
        }
    }
};

// A result record for the polling host thread (Tracker::track_wait) and for the chained launch of the next frame: two
// naturally aligned 8-byte granules {x, seq} and {y, seq << 2 | status}, each validated by its own tag, written by one
// 16-byte store to fine-grained host memory.  Nothing in HIP or PCIe promises that a 16-byte store arrives as one
// indivisible write, so the reader checks the tag of each half: a torn record cannot pair a new tag with stale data
// (aligned 8-byte granules written by one store are the hand-off unit of /opt/skills/guides/MI355X_MICROARCH.md,
// "handoff-1to1").  No fence and no wait: a system-scope release would write back the XCD's whole L2 200 times per
// frame under the warp kernel that merges partial output lines there (warp beside the tracker 72 us against 52), and
// four ordered stores with a wait for their acknowledgement cost the tracker chain a third of the pipeline's rate.
__device__ __forceinline__ uint4 make_record(float x, float y, unsigned int status, unsigned int seq) {
    return make_uint4(__float_as_uint(x), seq, __float_as_uint(y), (seq << 2) | status);
}
__device__ __forceinline__ unsigned int record_status(const uint4 &r) { return r.w & 3u; }

// The device copy of a record is what the NEXT launch starts from, slot by slot.  A chained launch sits on the SAME stream
// behind its parent, so the record is complete when the child starts: a plain 16-byte store here, a plain load there, and the
// child checks the tag (a mismatch means the host chained the wrong buffers: reported as status 3, never tracked from).
#ifdef VSTAB_DEV
// development builds: sixteen 100 MHz wall-clock stamps per feature and launch (tools/lk_timeline.py):
// [0] entry, [1] start point known, [2] blocks staged, then per level (3 -> 0): [3 + 3 i] derivatives + patch matrix done,
// [4 + 3 i] iterations done, [5 + 3 i] iteration count; [15] end
__device__ unsigned long long *g_lk_timing = nullptr;
extern "C" __attribute__((visibility("default"))) void vstab_dev_set_lk_timing(void *p) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lk_timing), &p, sizeof(p));
}
#define LK_STAMP(k, v) \
    if (g_lk_timing && tid == 0) g_lk_timing[(((size_t)(seq & 63u) * 256 + (size_t)f) << 5) + (k)] = (v)
#define LK_STAMP1(k, v) \
    if (g_lk_timing && tid == 64) g_lk_timing[(((size_t)(seq & 63u) * 256 + (size_t)f) << 5) + (k)] = (v)
#define LK_NOW() __builtin_amdgcn_s_memrealtime()
#else
#define LK_STAMP(k, v)
#define LK_STAMP1(k, v)
#define LK_NOW() 0ull
#endif

// One launch tracks every feature slot through args.n_frames CONSECUTIVE frame pairs (a "segment"): frame pair i is
// (args.pyr[i], args.pyr[i + 1]); a slot starts pair 0 from the host's point list (prev_pts) or from the record the parent
// launch left for it (chain_in), and pair i + 1 from the point it tracked to in pair i -- FrameSourceWarp.cpp:427, where the
// surviving points of one frame are the next frame's input.  Every slot runs down its own chain at its own pace: the launch
// lasts as long as the slowest slot's SUM over the frames instead of the sum over the frames of each frame's slowest slot, and
// the gap between launches is paid once per segment.  Results leave per frame pair (host_rec[i], dev_rec[i]) as soon as the
// slot has them.  A slot that loses its feature reports status 0 for that pair and status 2 ("lost earlier") for the rest.
__global__ void __launch_bounds__(LK_THREADS, 3) k_lk_track(LkSegArgs args) {
    __shared__ __attribute__((aligned(16))) int regI[LK_MAX_LEVELS][LKR * LKR];
    __shared__ uint32_t dpk[LK_MAX_LEVELS][LKT * LKT];     // Scharr derivative pairs of the 22 x 22 taps: dx | dy << 16 (int16 each)
    __shared__ short patch_i[LK_MAX_LEVELS][LK_WPAD];      // the interpolated window of every level: I ...
    __shared__ uint32_t patch_xy[LK_MAX_LEVELS][LK_WPAD];  // ... and Ix | Iy << 16 (int16 each), as the iterations hold them in registers
    __shared__ float level_mat[LK_MAX_LEVELS][8];          // A11, A12, A22, 1 / D of the level's 2 x 2 matrix and its rejection flag (lk_level_matrix)
    __shared__ __attribute__((aligned(16))) int regJ[2][LKJR * LKJR];
    // (wave as a scalar: what is indexed with it -- the pyramid level a wave prepares -- is then read with scalar loads from the
    // kernel arguments instead of per-lane global loads, each of which was a memory latency inside the dependent chain)
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = args.n;
    if (f >= n) return;
    unsigned long long *const clk = args.clk;
    __builtin_amdgcn_s_setprio(VSTAB_LK_PRIO);  // with the warp, above the pyramid and detector kernels (vstab_warp_fused.hip: VSTAB_WARP_PRIO)
    // development aid (VSTAB_LK_CLOCK): first workgroup start / last workgroup end on the 100 MHz wall clock
    if (clk && tid == 0) atomicMin(&clk[0], wall_clock64());
    unsigned int seq = args.seq[0];
    LK_STAMP(0, LK_NOW());
    float2 pp;
    if (args.chain_in) {
        // chained launch: this slot's input is the record the parent launch wrote for it in its LAST frame pair -- the point it
        // tracked to, if it survived (status 1).  Slots that were lost earlier stay lost (status 2) and are skipped by the
        // host, which is exactly the status filter of FrameSourceWarp.cpp:261-268.
        const uint4 r = args.chain_in[f];
        const bool tagged = r.y == args.parent_seq && (r.w >> 2) == (args.parent_seq & 0x3fffffffu);
        if (!tagged || record_status(r) != 1u) {  // uniform for the workgroup
            if (tid == 0) {
                for (int i = 0; i < args.n_frames; i++) {
                    const uint4 dead = make_record(0.0f, 0.0f, tagged && record_status(r) != 3u ? 2u : 3u, args.seq[i]);
                    if (args.dev_rec[i]) args.dev_rec[i][f] = dead;
                    if (args.host_rec[i]) args.host_rec[i][f] = dead;
                }
            }
            return;
        }
        pp = make_float2(__uint_as_float(r.x), __uint_as_float(r.z));
    } else {
        pp = args.prev_pts[f];
    }
    const float half = (LKW - 1) * 0.5f;
    // The Gauss-Newton iterations of a level run on ONE wave (wave 0), seven window pixels per lane: k = lane + 64 m (441 pixels;
    // the last 7 lanes have six).  With the window spread over four waves (two pixels per lane) an iteration was ~300 instructions of
    // splitting, reducing, exchanging through LDS and a barrier around eight tap reads -- and a wave alone on its SIMD issues a
    // dependent instruction every ~8 cycles whatever it does (the measured 0.8 - 1.1 us per iteration).  Seven independent pixels per
    // lane pipeline at the issue rate, one DPP reduction gives the total, nothing is exchanged and nobody waits at a barrier; the
    // other three waves stage the next level's block and sleep at the level's barrier.  The sums are exact integers either way.
    constexpr int LK_PPL = (LKW * LKW + 63) / 64;  // 7
    int koff[LK_PPL];  // offset of window pixel k inside a staged next-image block (a pixel the lane does not have: 0, with a zero patch)
#pragma unroll
    for (int m = 0; m < LK_PPL; m++) {
        const int k = lane + 64 * m;
        const int wy = lk_div21(k), wx = k - lk_mul(wy, LKW);
        koff[m] = k < LKW * LKW ? lk_mul(wy, LKJR) + wx : 0;
    }
    __shared__ float s_result[4];  // wave 0 -> all: the point the frame pair ended on and its status
    __shared__ float s_est[2][2];  // wave 0 -> all, by level parity: where the feature stands in the next image after that level
    // Fetched AHEAD, during the last level of a frame pair, for the NEXT pair of the segment -- by waves 1 - 3, in the shadow of wave 0's
    // iterations; the dwords wait in registers: (a) the previous-image neighbourhoods of all levels -- blocks of the current next image
    // around where the feature stands now, wide enough (32 x 32 for a 24 x 24 neighbourhood) to hold the neighbourhood of the point the
    // pair finally ends on: the top level's by the three waves together (it is needed first), every lower level's by the wave that will
    // prepare that level (it alone writes the neighbourhood to LDS, right before it reads it) -- and (b) the TOP level's next-image
    // block around the position PREDICTED there: this pair's end plus this pair's motion.  Without them a pair began with one exposed
    // memory latency (2.2 - 2.9 of a feature's 18 us in the 4K pipeline).  A block that turns out not to hold what is needed -- the
    // motion changed by more than the slack, or the block would cross the image border (blocks fetched ahead are loaded without border
    // handling) -- is fetched as before.  Where a block sits never changes a value: it holds image bytes either way.  The origins
    // travel through LDS (s_pf), so that wave 0 spends no instruction on any of it.  (The lower levels' next-image blocks fetched ahead
    // the same way were measured and dropped: the fetch one level ahead already hides their latency, and a predicted block is left by
    // the Gauss-Newton window more often than one centred on the running estimate.)
    LkStage<LKJR, LK_THREADS - 64> pfIt, pfJt;
    LkStage<LKJR, 64> pfIo;
    constexpr int PF_NONE = INT_MIN / 2;
    constexpr int PF_SLACK = (LKJR - LKR) / 2;  // 4 pixels each side
    __shared__ int s_pf[LK_MAX_LEVELS][4];      // origins of the blocks fetched ahead: neighbourhood x, y, next-image block x, y (PF_NONE: no block)
    __shared__ int s_jorg[2][2];                // waves 1 - 3 -> wave 0, by buffer: origin of the next-image block staged for the coming level
#pragma unroll 1
    for (int fi = 0; fi < args.n_frames; fi++) {
    const LkPyramid &I = args.pyr[fi], &J = args.pyr[fi + 1];
    seq = args.seq[fi];
    if (fi) LK_STAMP(0, LK_NOW());
    LK_STAMP(1, LK_NOW());
    float2 np = make_float2(0.f, 0.f);
    int st = 1;
    const int max_level = I.levels - 1;
    // What was not fetched ahead (the first pair of a launch; a block that does not hold what is needed) is loaded now, in one
    // exposed latency: the previous-image neighbourhood of EVERY level (it depends on the feature position only) and the top
    // level's next-image block.  The block of each lower level is fetched one level ahead, while the level above computes, around
    // the position the feature is expected at; if the Gauss-Newton window ends up outside it, the block is staged again around
    // the window.
    int jb = 0;                                   // regJ[jb] holds the block staged for the level about to run
    int jorg_x = PF_NONE, jorg_y = PF_NONE;       // its origin (none yet)
    int own_dx = -1, own_dy = -1;  // waves 1 - 3: the neighbourhood of the level this wave prepares lies in pfIo, at this offset
    {
        LkStage<LKR> si[LK_MAX_LEVELS];
        LkStage<LKJR> sj;
        bool iok[LK_MAX_LEVELS];  // the level's neighbourhood is loaded now
        int top_dx = -1, top_dy = -1;
        bool top_ahead = false;
#pragma unroll
        for (int l = 0; l < LK_MAX_LEVELS; l++) {
            iok[l] = false;
            if (l > max_level) continue;
            int ax = PF_NONE, ay = PF_NONE;
            if (fi) ax = __builtin_amdgcn_readfirstlane(s_pf[l][0]), ay = __builtin_amdgcn_readfirstlane(s_pf[l][1]);  // (published by the previous pair's last level)
            const float lscale = lk_level_scale(l);
            const int ipx = (int)floorf(pp.x * lscale - half), ipy = (int)floorf(pp.y * lscale - half);
            if (ipx < -LKW || ipx >= I.w[l] || ipy < -LKW || ipy >= I.h[l]) continue;  // the level loop skips it too
            const int dx = ipx - 1 - ax, dy = ipy - 1 - ay;  // (huge without a block)
            if ((unsigned)dx <= (unsigned)(2 * PF_SLACK) && (unsigned)dy <= (unsigned)(2 * PF_SLACK)) {
                if (l == max_level) top_dx = dx, top_dy = dy;
                else if (l == max_level - wave) own_dx = dx, own_dy = dy;
            } else {
                iok[l] = true;
                si[l].load(I.img[l], (uint32_t)I.pitch[l], I.w[l], I.h[l], ipx - 1, ipy - 1, tid);
            }
            if (l == max_level) {  // the top level starts at the feature position itself
                int bx = PF_NONE, by = PF_NONE;
                if (fi) bx = __builtin_amdgcn_readfirstlane(s_pf[l][2]), by = __builtin_amdgcn_readfirstlane(s_pf[l][3]);
                if (ipx - 2 >= bx && ipy - 2 >= by && ipx + LKT + 2 <= bx + LKJR && ipy + LKT + 2 <= by + LKJR) {
                    jorg_x = bx, jorg_y = by, top_ahead = true;
                } else {
                    jorg_x = ipx - LKJM, jorg_y = ipy - LKJM;
                    sj.load(J.img[l], (uint32_t)J.pitch[l], I.w[l], I.h[l], jorg_x, jorg_y, tid);
                }
            }
        }
#pragma unroll
        for (int l = 0; l < LK_MAX_LEVELS; l++)
            if (iok[l]) si[l].store(regI[l], tid);
        if (top_ahead && wave != 0) pfJt.store(regJ[0], tid - 64);
        if (top_dx >= 0 && wave != 0) pfIt.template store_window<LKR>(regI[max_level], tid - 64, top_dx, top_dy);
        if (jorg_x != PF_NONE && !top_ahead) sj.store(regJ[0], tid);
        LK_STAMP(17, (unsigned long long)((top_dx >= 0 ? 1 : 0) | (top_ahead ? 2 : 0)));
    }
    LK_STAMP(2, LK_NOW());
    LK_STAMP(16, (unsigned long long)fi);
    // ---- the previous image's side of every level: derivatives, interpolated window (I, Ix, Iy), sums of the 2 x 2 matrix ---------
    // They depend on the feature's position in the previous image only -- not on anything the Gauss-Newton iterations produce.
    // The TOP level is needed first: all four waves prepare it together (two derivative taps and two window pixels per lane, the
    // matrix sums joined through LDS), which takes a quarter of the time one wave needs for a level.  Then wave 0 starts iterating
    // on it while waves 1 .. 3 prepare the lower levels, one level each, in its shadow (inside the level loop below) -- before,
    // every level was prepared up front, one wave per level, and the top level's iterations waited for all of them (3.8 of a
    // feature's ~19 us).  The sums are exact integers whatever the decomposition, so every float derived from them is unchanged.
    struct LevelGeom {
        int ipx, ipy, iw00, iw01, iw10, iw11;
        bool ok;
    };
    auto level_geom = [&](int l) {
        const float ls = lk_level_scale(l);
        float qx = pp.x * ls, qy = pp.y * ls;
        qx -= half, qy -= half;
        LevelGeom g;
        g.ipx = (int)floorf(qx), g.ipy = (int)floorf(qy);
        g.ok = l >= 0 && l <= max_level && !(g.ipx < -LKW || g.ipx >= I.w[l] || g.ipy < -LKW || g.ipy >= I.h[l]);  // the level loop skips it too
        const float a = qx - (float)g.ipx, b = qy - (float)g.ipy;
        g.iw00 = (int)rintf((1.f - a) * (1.f - b) * 16384.f);
        g.iw01 = (int)rintf(a * (1.f - b) * 16384.f);
        g.iw10 = (int)rintf((1.f - a) * b * 16384.f);
        g.iw11 = 16384 - g.iw00 - g.iw01 - g.iw10;
        return g;
    };
    // Scharr derivative pairs of taps e = first, first + stride, ... of level l's 22 x 22 tap block (zero outside the image)
    auto level_derivatives = [&](int l, const LevelGeom &g, int first, int stride) {
        const int *rI = regI[l];
        const int w = I.w[l], h = I.h[l];
        for (int e = first; e < LKT * LKT; e += stride) {
            const int tyy = lk_div22(e), txx = e - lk_mul(tyy, LKT);
            const int X = g.ipx + txx, Y = g.ipy + tyy;
            int dx = 0, dy = 0;
            if (X >= 0 && Y >= 0 && X < w && Y < h) {
                const int *c = &rI[lk_mul(tyy + 1, LKR) + (txx + 1)];
                const int t0m = lk_mad(c[-1], 10, lk_mad(c[-LKR - 1] + c[LKR - 1], 3, 0)), t0p = lk_mad(c[1], 10, lk_mad(c[-LKR + 1] + c[LKR + 1], 3, 0));
                const int t1m = c[LKR - 1] - c[-LKR - 1], t1c = c[LKR] - c[-LKR], t1p = c[LKR + 1] - c[-LKR + 1];
                dx = (short)(t0p - t0m), dy = (short)lk_mad(t1c, 10, lk_mad(t1p + t1m, 3, 0));
            }
            dpk[l][e] = ((uint32_t)dx & 0xffffu) | ((uint32_t)dy << 16);
        }
    };
    // window pixels k = first, first + stride, ... of level l -> patch[l]; this lane's share of the matrix sums -> t (hi / lo parts)
    auto level_window = [&](int l, const LevelGeom &g, int first, int stride, int (&t)[6]) {
        const int *rI = regI[l];
        int pA[3] = {0, 0, 0};  // per-lane partial sums: 7 * 4080^2 < 2^27
        for (int k = first; k < LKW * LKW; k += stride) {
            const int wy = lk_div21(k), wx = k - lk_mul(wy, LKW);
            const int *c = &rI[lk_mul(wy + 1, LKR) + (wx + 1)];
            const int ival = LK_DESCALE(lk_mul(c[0], g.iw00) + lk_mul(c[1], g.iw01) + lk_mul(c[LKR], g.iw10) + lk_mul(c[LKR + 1], g.iw11), 9);
            const uint32_t *d = &dpk[l][lk_mul(wy, LKT) + wx];
            const uint32_t d00 = d[0], d01 = d[1], d10 = d[LKT], d11 = d[LKT + 1];
            const int ixval = LK_DESCALE(lk_mul((int)(short)(d00 & 0xffffu), g.iw00) + lk_mul((int)(short)(d01 & 0xffffu), g.iw01) +
                                             lk_mul((int)(short)(d10 & 0xffffu), g.iw10) + lk_mul((int)(short)(d11 & 0xffffu), g.iw11), 14);
            const int iyval = LK_DESCALE(lk_mul((int)d00 >> 16, g.iw00) + lk_mul((int)d01 >> 16, g.iw01) + lk_mul((int)d10 >> 16, g.iw10) +
                                             lk_mul((int)d11 >> 16, g.iw11), 14);
            patch_i[l][k] = (short)ival, patch_xy[l][k] = ((uint32_t)ixval & 0xffffu) | ((uint32_t)iyval << 16);
            pA[0] += lk_mul(ixval, ixval), pA[1] += lk_mul(ixval, iyval), pA[2] += lk_mul(iyval, iyval);
        }
#pragma unroll
        for (int q = 0; q < 3; q++) t[2 * q] = pA[q] >> 16, t[2 * q + 1] = pA[q] & 0xffff;
        wave_sums_i32(t);  // uniform; hi * 65536 + lo is the exact total of this wave's pixels
    };
    __shared__ int s_top[LK_WAVES][6];  // the waves' shares of the top level's matrix sums
    static_assert(LK_MAX_LEVELS <= LK_WAVES, "wave 0 iterates, waves 1 .. prepare one lower level each");
    {
        const LevelGeom gt = level_geom(max_level);
        __syncthreads();  // the staged neighbourhoods are visible
        if (gt.ok) level_derivatives(max_level, gt, tid, LK_THREADS);
        __syncthreads();  // the top level's derivative pairs are visible to every wave
        int t[6] = {0, 0, 0, 0, 0, 0};
        if (gt.ok) level_window(max_level, gt, tid, LK_THREADS, t);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 6; i++) s_top[wave][i] = t[i];
        }
        __syncthreads();  // the top level's window and sums, the block staged for it: wave 0 can start iterating
        if (tid == 0) {
            float sums[3];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                int hi = 0, lo = 0;
#pragma unroll
                for (int wv = 0; wv < LK_WAVES; wv++) hi += s_top[wv][2 * q], lo += s_top[wv][2 * q + 1];
                sums[q] = (float)__builtin_fma((double)hi, 65536.0, (double)lo);
            }
            lk_level_matrix(sums[0], sums[1], sums[2], level_mat[max_level]);  // (read back by this wave only)
        }
    }
    for (int level = max_level; level >= 0; level--) {
        int n_iter = 0;
        const uint8_t *jmg = J.img[level];
        const int w = I.w[level], h = I.h[level];
        const uint32_t jpitch = (uint32_t)J.pitch[level];
        const float lscale = lk_level_scale(level);
        // the block staged for this level, wave 0's estimate after the level above and the level's window are visible (the top level:
        // behind the barrier above)
        if (level != max_level) __syncthreads();
        LK_STAMP(18 + 3 * (max_level - level), LK_NOW());
        if (wave != 0) {
            // ---- waves 1 - 3: everything around the iterations, so that the iterating wave spends no instruction on it and never waits
            // for global memory.
            // The next (finer) level's next-image block; it lands while this level iterates.  It is centred on where the feature is
            // EXPECTED there -- the estimate this level starts from, doubled -- not on the zero-flow position: with the block around the
            // previous frame's position a frame-to-frame motion of more than LKJM pixels at that level (the 4K bench clip moves 4 - 12)
            // meant staging the block again inside the iteration loop, a global-memory latency on the dependent chain.  The block fetched
            // ahead for that level (pfJ) serves if it holds the expected window with two pixels to spare.  A window that leaves its block
            // is staged afresh by wave 0.
            LkStage<LKJR, LK_THREADS - 64> sn;
            int nx0 = PF_NONE, ny0 = PF_NONE;
            if (level > 0) {
                const int w1 = I.w[level - 1], h1 = I.h[level - 1];
                const float ex = level == max_level ? pp.x * lscale : s_est[(level + 1) & 1][0] * 2.0f, ey = level == max_level ? pp.y * lscale : s_est[(level + 1) & 1][1] * 2.0f;
                const float cx = ex * 2.0f - half, cy = ey * 2.0f - half;
                if (cx > -(float)(LKW + 1) && cx < (float)w1 && cy > -(float)(LKW + 1) && cy < (float)h1) {  // (false for NaN: a diverged estimate is not chased)
                    nx0 = __builtin_amdgcn_readfirstlane((int)floorf(cx)) - LKJM, ny0 = __builtin_amdgcn_readfirstlane((int)floorf(cy)) - LKJM;
                    sn.load(J.img[level - 1], (uint32_t)J.pitch[level - 1], w1, h1, nx0, ny0, tid - 64);
                }
                if (tid == 64) s_jorg[jb ^ 1][0] = nx0, s_jorg[jb ^ 1][1] = ny0;
            } else {
                // the last level of the pair: fetch ahead for the next pair (pfIt / pfIo / pfJ above).  Where the feature stands now: level
                // 1's result, doubled; its motion in this pair: that minus the pair's start point.
                const bool ahead = fi + 1 < args.n_frames && max_level >= 1;  // uniform
                const LkPyramid &J2 = args.pyr[ahead ? fi + 2 : fi + 1];
                const float e0x = s_est[1][0] * 2.0f, e0y = s_est[1][1] * 2.0f;
                const float p0x = e0x + (e0x - pp.x), p0y = e0y + (e0y - pp.y);
                int oix[LK_MAX_LEVELS], oiy[LK_MAX_LEVELS], ojx[LK_MAX_LEVELS], ojy[LK_MAX_LEVELS];
#pragma unroll
                for (int l = LK_MAX_LEVELS - 1; l >= 0; l--) {  // needed first: the neighbourhoods in what is now the next image, top level first
                    oix[l] = oiy[l] = PF_NONE;
                    if (!ahead || l > max_level) continue;
                    const float ls = lk_level_scale(l);
                    const float cx = e0x * ls - half, cy = e0y * ls - half;
                    if (cx > -(float)(LKW + 1) && cx < (float)I.w[l] && cy > -(float)(LKW + 1) && cy < (float)I.h[l]) {
                        const int ox = __builtin_amdgcn_readfirstlane((int)floorf(cx)) - 1 - PF_SLACK, oy = __builtin_amdgcn_readfirstlane((int)floorf(cy)) - 1 - PF_SLACK;
                        if (ox >= 0 && oy >= 0 && ox + LKJR <= I.w[l] && oy + LKJR <= I.h[l]) {  // (a block across the image border is left to the pair's own staging)
                            oix[l] = ox, oiy[l] = oy;
                            if (l == max_level) pfIt.load_interior(J.img[l], (uint32_t)J.pitch[l], ox, oy, tid - 64);
                            else if (l == max_level - wave) pfIo.load_interior(J.img[l], (uint32_t)J.pitch[l], ox, oy, lane);
                        }
                    }
                }
#pragma unroll
                for (int l = 0; l < LK_MAX_LEVELS; l++) ojx[l] = ojy[l] = PF_NONE;
                if (ahead) {  // then the top-level block of the image after it, around the predicted position
                    const float ls = lk_level_scale(max_level);
                    const float cx = p0x * ls - half, cy = p0y * ls - half;
                    const int wt = I.w[max_level], ht = I.h[max_level];
                    if (cx > -(float)(LKW + 1) && cx < (float)wt && cy > -(float)(LKW + 1) && cy < (float)ht) {
                        const int ox = __builtin_amdgcn_readfirstlane((int)floorf(cx)) - LKJM, oy = __builtin_amdgcn_readfirstlane((int)floorf(cy)) - LKJM;
                        if (ox >= 0 && oy >= 0 && ox + LKJR <= wt && oy + LKJR <= ht) {
#pragma unroll
                            for (int l = 0; l < LK_MAX_LEVELS; l++)
                                if (l == max_level) ojx[l] = ox, ojy[l] = oy;
                            pfJt.load_interior(J2.img[max_level], (uint32_t)J2.pitch[max_level], ox, oy, tid - 64);
                        }
                    }
                }
                if (tid == 64) {
#pragma unroll
                    for (int l = 0; l < LK_MAX_LEVELS; l++) s_pf[l][0] = oix[l], s_pf[l][1] = oiy[l], s_pf[l][2] = ojx[l], s_pf[l][3] = ojy[l];
                }
            }
            if (level == max_level && wave <= max_level) {
                // in the shadow of the top level's iterations: wave w prepares level max_level - w on its own (a wave reads back only
                // what it wrote itself: LDS operations of one wave execute in order); published by that level's barrier
                const int l = max_level - wave;
                const LevelGeom g = level_geom(l);
                if (g.ok) {
                    if (own_dx >= 0) pfIo.template store_window<LKR>(regI[l], lane, own_dx, own_dy);  // fetched ahead by this wave
                    level_derivatives(l, g, lane, 64);
                    int t[6];
                    level_window(l, g, lane, 64, t);
                    if (lane == 0)
                        lk_level_matrix((float)__builtin_fma((double)t[0], 65536.0, (double)t[1]), (float)__builtin_fma((double)t[2], 65536.0, (double)t[3]),
                                        (float)__builtin_fma((double)t[4], 65536.0, (double)t[5]), level_mat[l]);
                }
            }
            // the finer level's block goes into the other buffer: nobody reads that one now (wave 0 left it before this level's
            // barrier); the next barrier publishes it
            if (nx0 != PF_NONE) sn.store(regJ[jb ^ 1], tid - 64);
        }
        if (wave == 0 && level != max_level) jorg_x = s_jorg[jb][0], jorg_y = s_jorg[jb][1];  // (published by this level's barrier)
        LK_STAMP(19 + 3 * (max_level - level), LK_NOW());
        if (wave == 0) do {  // one pyramid level on one wave ("break" = the reference's "continue")
            float ppx = pp.x * lscale, ppy = pp.y * lscale;
            float npx, npy;
            if (level == max_level)
                npx = ppx, npy = ppy;
            else
                npx = np.x * 2.0f, npy = np.y * 2.0f;
            np = make_float2(npx, npy);
            ppx -= half, ppy -= half;
            const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
            if (ipx < -LKW || ipx >= w || ipy < -LKW || ipy >= h) {
                if (level == 0) st = 0;
                break;
            }
            int *rJ = regJ[jb];
            int jx0 = jorg_x, jy0 = jorg_y;  // origin of the next-image block staged for this level
            int Iw[LK_PPL], Ixy[LK_PPL];     // this lane's pixels of the interpolated window: I, and Ix | Iy << 16
#pragma unroll
            for (int m = 0; m < LK_PPL; m++) {
                const int k = lane + 64 * m;
                const bool have = k < LKW * LKW;  // (the pad entries of the last row of lanes are never written)
                Iw[m] = have ? (int)patch_i[level][k] : 0;
                Ixy[m] = have ? (int)patch_xy[level][k] : 0;
            }
            const float FLT_SCALE = 1.0f / (1 << 20);
            const float A11 = level_mat[level][0], A12 = level_mat[level][1], A22 = level_mat[level][2], D = level_mat[level][3];
            if (level_mat[level][4] != 0.0f) {  // minEig < 1e-4 || D < FLT_EPSILON
                if (level == 0) st = 0;
                break;
            }
            npx -= half, npy -= half;
            float pdx = 0.f, pdy = 0.f;
            LK_STAMP(3 + 3 * (max_level - level), LK_NOW());
            for (int j = 0; j < 30; j++) {
                n_iter++;
                const int inx = (int)floorf(npx), iny = (int)floorf(npy);
                if (inx < -LKW || inx >= w || iny < -LKW || iny >= h) {
                    if (level == 0) st = 0;
                    break;
                }
                const float a = npx - (float)inx, b = npy - (float)iny;
                const int iw00 = (int)rintf((1.f - a) * (1.f - b) * 16384.f);
                const int iw01 = (int)rintf(a * (1.f - b) * 16384.f);
                const int iw10 = (int)rintf((1.f - a) * b * 16384.f);
                const int iw11 = 16384 - iw00 - iw01 - iw10;
                if (inx < jx0 || iny < jy0 || inx + LKT > jx0 + LKJR || iny + LKT > jy0 + LKJR) {
                    // the window is outside the staged block: this wave stages a block centred on the window (wave-uniform branch;
                    // LDS operations of one wave execute in order, and the other waves write the OTHER buffer)
                    jx0 = inx - LKJM, jy0 = iny - LKJM;
                    LkStage<LKJR, 64> sj;
                    sj.load(jmg, jpitch, w, h, jx0, jy0, lane);
                    sj.store(rJ, lane);
                }
                const int jbase = lk_mul(iny - jy0, LKJR) + (inx - jx0);
                int pb0 = 0, pb1 = 0;  // per-lane partial sums: 7 * 16320 * 4080 < 2^29
#pragma unroll
                for (int m = 0; m < LK_PPL; m++) {
                    const int *c = &rJ[jbase + koff[m]];
                    const int diff = LK_DESCALE(lk_mul(c[0], iw00) + lk_mul(c[1], iw01) + lk_mul(c[LKJR], iw10) + lk_mul(c[LKJR + 1], iw11), 9) - Iw[m];
                    pb0 += lk_mul(diff, (int)(short)(Ixy[m] & 0xffff)), pb1 += lk_mul(diff, Ixy[m] >> 16);
                }
                // exact totals: signed high part and 16-bit low part summed separately (each stays inside int32), joined in double
                int t[4] = {pb0 >> 16, pb0 & 0xffff, pb1 >> 16, pb1 & 0xffff};
                wave_sums_i32(t);  // uniform (SGPR) results
                const float sb0 = (float)__builtin_fma((double)t[0], 65536.0, (double)t[1]), sb1 = (float)__builtin_fma((double)t[2], 65536.0, (double)t[3]);
                const float b1 = sb0 * FLT_SCALE, b2 = sb1 * FLT_SCALE;
                const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
                npx += dx, npy += dy;
                np = make_float2(npx + half, npy + half);
                if ((double)dx * dx + (double)dy * dy <= 0.01 * 0.01) break;
                if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                    np.x -= dx * 0.5f, np.y -= dy * 0.5f;
                    break;
                }
                pdx = dx, pdy = dy;
            }
            // OpenCV's LKTrackerInvoker, behind the loop (the reference passes `err`, FrameSourceWarp.cpp:250-259): at
            // level 0 the final position -- the stored point minus the half window -- is tested against the image once
            // more, and a feature whose last step (or half-step correction) carried its window out is dropped.
            if (level == 0 && st) {
                const int fnx = (int)floorf(np.x - half), fny = (int)floorf(np.y - half);
                if (fnx < -LKW || fnx >= w || fny < -LKW || fny >= h) st = 0;
            }
        } while (false);
        if (tid == 0) s_est[level & 1][0] = np.x, s_est[level & 1][1] = np.y;  // (a skipped level leaves np at its start value, as the reference does)
        LK_STAMP(4 + 3 * (max_level - level), LK_NOW());
        LK_STAMP(5 + 3 * (max_level - level), (unsigned long long)n_iter);
        (void)n_iter;  // read by the development build's stamps only
        jb ^= 1;
    }
    // wave 0 holds the result: hand it to the other waves (the next frame pair starts from it; a lost slot ends here)
    if (tid == 0) s_result[0] = np.x, s_result[1] = np.y, s_result[2] = __int_as_float(st);
    __syncthreads();
    np = make_float2(s_result[0], s_result[1]), st = __float_as_int(s_result[2]);
    LK_STAMP(15, LK_NOW());
    if (tid == 0) {
        // one 16-byte record per feature and frame pair in coherent host memory (make_record); the host polls the tags.  The
        // device copy of the LAST pair feeds the launch chained behind this one.
        const uint4 rec = make_record(np.x, np.y, (unsigned int)st, seq);
        if (args.dev_rec[fi]) args.dev_rec[fi][f] = rec;
        if (args.host_rec[fi]) args.host_rec[fi][f] = rec;
        if (!st) {  // lost here: the remaining pairs of the segment report "lost earlier"
            for (int i = fi + 1; i < args.n_frames; i++) {
                const uint4 dead = make_record(0.0f, 0.0f, 2u, args.seq[i]);
                if (args.dev_rec[i]) args.dev_rec[i][f] = dead;
                if (args.host_rec[i]) args.host_rec[i][f] = dead;
            }
        }
        if (clk) atomicMax(&clk[1], wall_clock64());
    }
    if (!st) return;  // uniform
    pp = np;          // FrameSourceWarp.cpp:427
    __syncthreads();  // the last readers of this pair's LDS state are through before the next pair's staging overwrites it
    }  // frame pairs of the segment
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
// `done` (optional): an event that completes with this kernel -- bound to the launch itself (hipExtLaunchKernelGGL's stop event), so that the
// stream carries no marker packet of its own behind the kernel: a hipEventRecord after every frame's pyramid cost the prefetch stream ~6 us
// per frame (rocprofv3 kernel trace: the next frame's first kernel started 6.5 us after this frame's last one ended, 0.0 us between two kernels)
vstab_status launch_pyr_down(const uint8_t *src, size_t spitch, int sw, int sh, uint8_t *dst, size_t dpitch,
                             hipStream_t s, hipEvent_t done) {
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    // the kernel forms row offsets as 32-bit products
    if ((uint64_t)spitch * (uint64_t)sh >= (1ull << 32) || (uint64_t)dpitch * (uint64_t)dh >= (1ull << 32))
        return fail(VSTAB_ERR_INVALID, "pyr_down: planes of 4 GiB or more are not supported");
    const int vec_ok = reinterpret_cast<uintptr_t>(src) % 4 == 0 && spitch % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 4 == 0 && dpitch % 4 == 0;
    // group g = outputs 4g .. 4g+3 reads source bytes [8g - 4, 8g + 12): interior iff g >= 1, 8g + 12 <= sw and 4g + 4 <= dw
    const int n_groups = div_up(dw, 4);
    const int near = sw >= 16 && sh >= 4;
    const int g_lo = 1, g_hi = vec_ok && near ? std::max(g_lo, std::min(sw >= 12 ? (sw - 12) / 8 + 1 : 0, dw / 4)) : g_lo;
    const int nbx = div_up(g_hi - g_lo, 64), nb_int = nbx * div_up(dh, 4);
    const int nb_edge = div_up(dh * (n_groups - (g_hi - g_lo)), 256);
    hipExtLaunchKernelGGL(k_pyr_down, dim3(nb_edge + nb_int), dim3(256), 0, s, nullptr, done, 0, src, spitch, sw, sh, dst, dpitch, dw, dh, vec_ok, g_lo, g_hi, nbx, nb_edge,
                          n_groups, near);
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}

// Copy + first level in one launch (k_pack_pyr): ring = the NV12 frame (y, uv) packed (luma rows of pitch w, chroma rows behind them: what
// vstab_pack_nv12 writes), dst = pyrDown(luma).  Only where pack_pyr_ok says so; `copied` (optional) completes with the launch.
bool pack_pyr_ok(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int w, int h, const void *ring, const void *dst, size_t dpitch) {
    return w >= 16 && h >= 4 && !(w & 7) && !(h & 1) && reinterpret_cast<uintptr_t>(y) % 4 == 0 && pitch_y % 4 == 0 && reinterpret_cast<uintptr_t>(ring) % 8 == 0 &&
           reinterpret_cast<uintptr_t>(dst) % 4 == 0 && dpitch % 4 == 0 && pitch_y < (1u << 24) && pitch_uv < (1u << 24) && (uint64_t)pitch_y * (uint64_t)h < (1ull << 32) &&
           (uint64_t)pitch_uv * (uint64_t)(h / 2) < (1ull << 32) &&  // (k_pack_pyr forms row * pitch_uv in 32 bits)
           (uint64_t)w * (uint64_t)h * 3 / 2 < (1ull << 32);
}
vstab_status launch_pack_pyr(const uint8_t *y, size_t pitch_y, const uint8_t *uv, size_t pitch_uv, int sw, int sh, uint8_t *ring, uint8_t *dst, size_t dpitch,
                             hipStream_t s, hipEvent_t copied) {
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    if (!pack_pyr_ok(y, pitch_y, uv, pitch_uv, sw, sh, ring, dst, dpitch)) return fail(VSTAB_ERR_INVALID, "pack_pyr: planes not aligned for the fused copy");
    const int n_groups = div_up(dw, 4);
    const int g_lo = 1, g_hi = std::max(g_lo, std::min(sw >= 12 ? (sw - 12) / 8 + 1 : 0, dw / 4));
    const int nbx = div_up(g_hi - g_lo, 64), nb_int = nbx * div_up(dh, 4);
    const int nb_edge = div_up(dh * (n_groups - (g_hi - g_lo)), 256);
    const int uv_vec = reinterpret_cast<uintptr_t>(uv) % 16 == 0 && pitch_uv % 16 == 0 && sw % 16 == 0 && reinterpret_cast<uintptr_t>(ring) % 16 == 0;
    const int nb_uv = std::max(1, std::min(256, (int)div_up((unsigned)(sw * (sh / 2)), 256u * 16u)));
    hipExtLaunchKernelGGL(k_pack_pyr, dim3(nb_edge + nb_int + nb_uv), dim3(256), 0, s, nullptr, copied, 0, y, (uint32_t)pitch_y, sw, sh, dst, (uint32_t)dpitch, dw, dh, 1, g_lo,
                          g_hi, nbx, nb_edge, n_groups, 1, nb_edge + nb_int, uv, (uint32_t)pitch_uv, ring, (uint32_t)sw, uv_vec);
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}

// Two levels in one launch (k_pyr_down_x2): mid = pyrDown(src), dst = pyrDown(mid).  Needs an image a reflected tap never leaves
// twice (>= 4 x 4 at the middle level); the caller falls back to two single-level launches otherwise.
bool pyr_down_x2_ok(int sw, int sh) { return (sw + 1) / 2 >= 8 && (sh + 1) / 2 >= 8; }
vstab_status launch_pyr_down_x2(const uint8_t *src, size_t spitch, int sw, int sh, uint8_t *mid, size_t mpitch, uint8_t *dst, size_t dpitch, hipStream_t s,
                                hipEvent_t done) {
    const int mw = (sw + 1) / 2, mh = (sh + 1) / 2, dw = (mw + 1) / 2, dh = (mh + 1) / 2;
    if (!pyr_down_x2_ok(sw, sh)) return fail(VSTAB_ERR_INVALID, "pyr_down_x2: image too small");
    if ((uint64_t)spitch * (uint64_t)sh >= (1ull << 32) || (uint64_t)mpitch * (uint64_t)mh >= (1ull << 32) || (uint64_t)dpitch * (uint64_t)dh >= (1ull << 32))
        return fail(VSTAB_ERR_INVALID, "pyr_down_x2: planes of 4 GiB or more are not supported");
    const int vec_ok = reinterpret_cast<uintptr_t>(src) % 4 == 0 && spitch % 4 == 0 && reinterpret_cast<uintptr_t>(mid) % 4 == 0 && mpitch % 4 == 0;
    const int dst_vec_ok = reinterpret_cast<uintptr_t>(dst) % 4 == 0 && dpitch % 4 == 0;
    const int near = sw >= 16 && sh >= 4;
    hipExtLaunchKernelGGL(k_pyr_down_x2, dim3(div_up(dw, P2_TW), div_up(dh, P2_TH)), dim3(256), 0, s, nullptr, done, 0, src, (uint32_t)spitch, sw, sh, mid,
                          (uint32_t)mpitch, mw, mh, dst, (uint32_t)dpitch, dw, dh, vec_ok, dst_vec_ok, near);
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}

vstab_status launch_min_eig(const uint8_t *src, size_t pitch, int w, int h, float *eig, int *max_bits,
                            hipStream_t s) {
    VSTAB_HIP_TRY(hipMemsetAsync(max_bits, 0x80, sizeof(int), s));  // INT_MIN-ish (0x80808080): any value wins
    const int vec_ok = reinterpret_cast<uintptr_t>(src) % 4 == 0 && pitch % 4 == 0 && reinterpret_cast<uintptr_t>(eig) % 16 == 0;
    dim3 grid(div_up(w, ME_TW), div_up(h, ME_TH));
    hipLaunchKernelGGL(k_min_eig, grid, dim3(256), 0, s, src, pitch, w, h, eig, max_bits, vec_ok);
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}

vstab_status launch_corner_candidates(const float *eig, int w, int h, const int *max_bits, double quality,
                                      unsigned long long *keys, unsigned int *count, unsigned int cap,
                                      hipStream_t s) {
    VSTAB_HIP_TRY(hipMemsetAsync(count, 0, sizeof(unsigned int), s));
    dim3 grid(div_up(w, 64), div_up(h, 4));
    hipLaunchKernelGGL(k_corner_candidates, grid, dim3(64, 4), 0, s, eig, w, h, max_bits, quality, keys, count, cap);
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}

// Fused detector.  small: 8 dwords of device memory {biased maximum bits, -, -, -, keys kept, tiles that spilled, -, -};
// scratch: corners_fused_scratch_bytes(w, h) (per tile: 256 key slots, a count, and room for a dense 64 x 31 map that
// only a tile with more survivors than slots writes).  On return (stream order) keys holds min(small[4], cap) keys; the
// result is complete iff small[4] <= cap.
size_t corners_fused_scratch_bytes(int w, int h) {
    const size_t tiles = (size_t)div_up(w, CF_TW) * div_up(h, CF_TH);
    return tiles * (CF_SLOTS * sizeof(unsigned long long) + CF_TW * CF_TH * sizeof(float) + sizeof(unsigned int));
}

vstab_status launch_corners_fused(const uint8_t *src, size_t pitch, int w, int h, double quality, void *scratch, unsigned long long *keys,
                                  unsigned int cap, unsigned int *small, hipStream_t s) {
    VSTAB_HIP_TRY(hipMemsetAsync(small, 0, 8 * sizeof(unsigned int), s));
    const int vec_ok = reinterpret_cast<uintptr_t>(src) % 4 == 0 && pitch % 4 == 0;
    const int tiles_x = div_up(w, CF_TW), tiles_y = div_up(h, CF_TH), tiles = tiles_x * tiles_y;
    unsigned long long *slots = static_cast<unsigned long long *>(scratch);
    float *spill = reinterpret_cast<float *>(slots + (size_t)tiles * CF_SLOTS);
    unsigned int *tile_counts = reinterpret_cast<unsigned int *>(spill + (size_t)tiles * CF_TW * CF_TH);
    hipLaunchKernelGGL(k_corners_fused, dim3(tiles_x, tiles_y), dim3(256), 0, s, src, pitch, w, h, quality, small, slots, tile_counts, spill, vec_ok);
    hipLaunchKernelGGL(k_filter_keys, dim3(div_up(tiles, FK_TILES)), dim3(256), 0, s, slots, tile_counts, spill, tiles, tiles_x, w, small, quality, keys,
                       small + 4, cap);
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}

vstab_status launch_lk(const LkSegArgs &a, hipStream_t s) {
    if (a.n <= 0 || a.n_frames <= 0) return VSTAB_OK;
    if (a.n_frames > LK_SEG_MAX) return fail(VSTAB_ERR_INVALID, "launch_lk: too many frame pairs in one launch");
    if (!a.prev_pts && !a.chain_in) return fail(VSTAB_ERR_INVALID, "launch_lk: no input points");
    for (int i = 0; i <= a.n_frames; i++)
        for (int l = 0; l < a.pyr[i].levels; l++)
            if (a.pyr[i].pitch[l] >= (1u << 24) || (uint64_t)a.pyr[i].pitch[l] * (uint64_t)a.pyr[i].h[l] >= (1ull << 32))
                return fail(VSTAB_ERR_INVALID, "launch_lk: image pitch must be below 2^24 and planes below 4 GiB");
    hipLaunchKernelGGL(k_lk_track, dim3(a.n), dim3(LK_THREADS), 0, s, a);
    VSTAB_HIP_TRY(hipGetLastError());
    return VSTAB_OK;
}


// Kernels of this translation unit are one code object, loaded by the runtime at the first launch of any of them.  Touching one of them
// here (vstab_preload_kernels) moves that load to a moment the caller chooses.
vstab_status preload_track_kernels() {
    hipFuncAttributes at;
    VSTAB_HIP_TRY(hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&k_lk_track)));
    return VSTAB_OK;
}

}  // namespace vstab
