// vstab_internal.hpp -- host-side helpers shared by the translation units of libvstab.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>

#include "../../include/vstab.h"

namespace vstab {

void set_error(const std::string &msg);

inline vstab_status fail(vstab_status st, const std::string &msg) {
    set_error(msg);
    return st;
}

#define VSTAB_HIP_TRY(expr)                                                                     \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return ::vstab::fail(VSTAB_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

inline unsigned div_up(unsigned a, unsigned b) { return (a + b - 1) / b; }

// One-shot event pair for the NEXT hot-kernel launch on this thread: the launcher hands it to hipExtLaunchKernelGGL, which
// stamps the kernel's own start and end (what rocprofv3's kernel trace reports) instead of the stream positions around the
// launch call, whose interval also holds the dispatch wait behind other streams' kernels.  take_launch_events() clears it.
struct LaunchEvents {
    hipEvent_t start = nullptr, stop = nullptr;
};
void set_launch_events(hipEvent_t start, hipEvent_t stop);
LaunchEvents take_launch_events();

// one per translation unit with kernels: loads that unit's code object now (vstab_preload_kernels)
vstab_status preload_track_kernels();
vstab_status preload_warp_kernels();
vstab_status preload_fused_kernels();
vstab_status preload_p010_kernels();
vstab_status preload_planar_kernels();
bool launch_events_pending();

// vstab_pack_p010 with a choice of planes (vstab_warp.hip): luma_only narrows the luma plane alone -- what the 10-bit
// pipeline needs for its tracker
vstab_status pack_p010_planes(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int width, int height, void *dst, bool luma_only,
                              void *stream);

// vstab_pack_nv12 with an optional event that completes with the copy kernel (bound to the launch: no marker packet on the stream)
vstab_status pack_nv12_planes(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int width, int height, void *dst, void *stream, hipEvent_t done);

}  // namespace vstab
