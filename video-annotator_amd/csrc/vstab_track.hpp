// vstab_track.hpp -- launchers of the tracking kernels (vstab_track.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vstab.h"

namespace vstab {

constexpr int LK_MAX_LEVELS = 4;  // maxLevel 3 (calcOpticalFlowPyrLK default)
constexpr int LK_WIN = 21;

struct LkPyramid {
    const uint8_t *img[LK_MAX_LEVELS];
    size_t pitch[LK_MAX_LEVELS];
    int w[LK_MAX_LEVELS], h[LK_MAX_LEVELS];
    int levels;
};

// number of levels buildOpticalFlowPyramid produces for maxLevel 3, winSize 21
inline int lk_levels(int w, int h) {
    int n = 1;
    for (int l = 1; l < LK_MAX_LEVELS; l++) {
        w = (w + 1) / 2, h = (h + 1) / 2;
        if (w <= LK_WIN || h <= LK_WIN) break;
        n++;
    }
    return n;
}

// done (optional): an event that completes with the kernel, bound to the launch itself (no marker packet of its own on the stream)
vstab_status launch_pyr_down(const uint8_t *src, size_t spitch, int sw, int sh, uint8_t *dst, size_t dpitch, hipStream_t s, hipEvent_t done = nullptr);
// the copy of an NV12 frame into the ring (what vstab_pack_nv12 writes: luma rows of pitch w, chroma rows behind them) and the first pyramid
// level of its luma in one launch; `copied` (optional) completes with it
bool pack_pyr_ok(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int w, int h, const void *ring, const void *dst, size_t dpitch);
vstab_status launch_pack_pyr(const uint8_t *y, size_t pitch_y, const uint8_t *uv, size_t pitch_uv, int w, int h, uint8_t *ring, uint8_t *dst, size_t dpitch, hipStream_t s,
                             hipEvent_t copied = nullptr);
// two levels in one launch: mid = pyrDown(src) ((sw+1)/2 x (sh+1)/2), dst = pyrDown(mid); only where pyr_down_x2_ok(sw, sh)
bool pyr_down_x2_ok(int sw, int sh);
vstab_status launch_pyr_down_x2(const uint8_t *src, size_t spitch, int sw, int sh, uint8_t *mid, size_t mpitch, uint8_t *dst, size_t dpitch, hipStream_t s,
                                hipEvent_t done = nullptr);
vstab_status launch_min_eig(const uint8_t *src, size_t pitch, int w, int h, float *eig, int *max_bits, hipStream_t s);
vstab_status launch_corner_candidates(const float *eig, int w, int h, const int *max_bits, double quality,
                                      unsigned long long *keys, unsigned int *count, unsigned int cap, hipStream_t s);
size_t corners_fused_scratch_bytes(int w, int h);
vstab_status launch_corners_fused(const uint8_t *src, size_t pitch, int w, int h, double quality, void *scratch, unsigned long long *keys,
                                  unsigned int cap, unsigned int *small, hipStream_t s);
// One tracker launch = a SEGMENT of up to LK_SEG_MAX consecutive frame pairs (k_lk_track): pair i is (pyr[i], pyr[i + 1]).
//   prev_pts  the n start points of pair 0 (device-readable), or NULL when
//   chain_in  the device records of the parent launch's last pair, whose sequence number is parent_seq: slot f starts from the
//             point record f holds (status 1) or reports status 2 ("lost earlier") for every pair; a record with another tag
//             is a bookkeeping error (status 3).  The parent must precede this launch on the same stream.
//   host_rec[i] / dev_rec[i] (either may be NULL): where pair i's n 16-byte records {x, seq[i], y, seq[i] << 2 | status} go --
//             mapped host memory the host polls / device memory for the launch chained behind this one
//   clk       development aid (VSTAB_LK_CLOCK): first-workgroup start / last-workgroup end stamps
constexpr int LK_SEG_MAX = 8;
struct LkSegArgs {
    LkPyramid pyr[LK_SEG_MAX + 1];
    uint4 *host_rec[LK_SEG_MAX];
    uint4 *dev_rec[LK_SEG_MAX];
    unsigned int seq[LK_SEG_MAX];
    int n_frames, n;
    const float2 *prev_pts;
    const uint4 *chain_in;
    unsigned int parent_seq;
    unsigned long long *clk;
};
vstab_status launch_lk(const LkSegArgs &args, hipStream_t s);

}  // namespace vstab
