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

vstab_status launch_pyr_down(const uint8_t *src, size_t spitch, int sw, int sh, uint8_t *dst, size_t dpitch, hipStream_t s);
vstab_status launch_min_eig(const uint8_t *src, size_t pitch, int w, int h, float *eig, int *max_bits, hipStream_t s);
vstab_status launch_corner_candidates(const float *eig, int w, int h, const int *max_bits, double quality,
                                      unsigned long long *keys, unsigned int *count, unsigned int cap, hipStream_t s);
size_t corners_fused_scratch_bytes(int w, int h);
vstab_status launch_corners_fused(const uint8_t *src, size_t pitch, int w, int h, double quality, void *scratch, unsigned long long *keys,
                                  unsigned int cap, unsigned int *small, hipStream_t s);
// host_records (may be NULL): n 16-byte records {x, seq, y, seq << 2 | status} in mapped host memory, written instead
// of next_pts / status so the host can poll for completion without a stream synchronisation
// chain_in (may be NULL): the device records of the previous frame's launch, whose sequence number is parent_seq; slot f
// WAITS for that launch's record f (the two launches may run side by side on different streams), then starts from the
// point it holds (status 1) or reports status 2 ("lost earlier") without tracking; status 3 = the record never came.
// dev_records (may be NULL): device copy of the records for the launch chained behind this one.
vstab_status launch_lk(const LkPyramid &I, const LkPyramid &J, const float2 *prev_pts, int n, float2 *next_pts,
                       uint8_t *status, void *host_records, unsigned int seq, hipStream_t s, const void *chain_in = nullptr,
                       unsigned int parent_seq = 0, void *dev_records = nullptr, void *clock_pair = nullptr);

}  // namespace vstab
