// ref_cl_runner.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Runs the reference's own createMap kernel on the GPU: oracle/_ref/createMap.gfx950.co is
// /root/reference/opencv/createMap.cl compiled UNMODIFIED by ROCm's OpenCL front end for gfx950
// (recipe: oracle/Makefile, target `ref_gfx950`; ROCm's own ocml/ockl/opencl bitcode, no stand-ins).
// The launch mirrors FrameSourceWarp.cpp:272-304: global size {cols, rows}, explicit arguments in
// the order of :275-300 (KernelArg::WriteOnly = ptr, step, offset, rows, cols; WriteOnlyNoSize =
// ptr, step, offset; then the 17 cl_float values).  The hidden arguments the code object's metadata
// lists (block counts, group sizes, global offsets) are filled in by the HIP runtime, which shares
// its kernel-argument machinery with the OpenCL runtime.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

namespace {

struct __attribute__((packed)) CreateMapArgs {
    void *map_x;
    int step_x, offset_x, rows, cols;
    void *map_y;
    int step_y, offset_y;
    float f[17];
};
static_assert(sizeof(CreateMapArgs) == 108, "explicit kernarg block of createMap (metadata: 0..107)");

int fail(char *err, int errlen, const char *what, hipError_t e) {
    if (err && errlen > 0) snprintf(err, (size_t)errlen, "%s: %s", what, hipGetErrorString(e));
    return -1;
}

}  // namespace

#define RC_TRY(expr)                                          \
    do {                                                      \
        hipError_t e_ = (expr);                               \
        if (e_ != hipSuccess) {                               \
            if (mod) (void)hipModuleUnload(mod);              \
            if (dx) (void)hipFree(dx);                        \
            if (dy) (void)hipFree(dy);                        \
            return fail(err, errlen, #expr, e_);              \
        }                                                     \
    } while (0)

// Map planes of the reference kernel for a cols x rows output, copied to host memory (dense, row-major).
// block_x / block_y: the work-group size (OpenCV passes NULL = runtime's choice; the result does not depend on it).
extern "C" __attribute__((visibility("default"))) int refcl_create_map(const char *co_path, int cols, int rows, const float params[17],
                                                                       float *out_x, float *out_y, int block_x, int block_y, char *err,
                                                                       int errlen) {
    hipModule_t mod = nullptr;
    void *dx = nullptr, *dy = nullptr;
    if (cols <= 0 || rows <= 0 || cols > 32767 || rows > 32767 || block_x <= 0 || block_y <= 0 || block_x * block_y > 1024) {
        if (err && errlen > 0) snprintf(err, (size_t)errlen, "bad size");
        return -1;
    }
    RC_TRY(hipModuleLoad(&mod, co_path));
    hipFunction_t fn = nullptr;
    RC_TRY(hipModuleGetFunction(&fn, mod, "createMap"));
    const size_t bytes = (size_t)cols * rows * sizeof(float);
    RC_TRY(hipMalloc(&dx, bytes));
    RC_TRY(hipMalloc(&dy, bytes));
    RC_TRY(hipMemset(dx, 0xff, bytes));  // NaN pattern: a pixel the kernel does not write is noticed
    RC_TRY(hipMemset(dy, 0xff, bytes));
    CreateMapArgs a;
    a.map_x = dx, a.step_x = cols * 4, a.offset_x = 0, a.rows = rows, a.cols = cols;
    a.map_y = dy, a.step_y = cols * 4, a.offset_y = 0;
    memcpy(a.f, params, sizeof(a.f));
    size_t size = sizeof(a);
    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    const unsigned gx = (unsigned)((cols + block_x - 1) / block_x), gy = (unsigned)((rows + block_y - 1) / block_y);
    RC_TRY(hipModuleLaunchKernel(fn, gx, gy, 1, (unsigned)block_x, (unsigned)block_y, 1, 0, nullptr, nullptr, extra));
    RC_TRY(hipDeviceSynchronize());
    RC_TRY(hipMemcpy(out_x, dx, bytes, hipMemcpyDeviceToHost));
    RC_TRY(hipMemcpy(out_y, dy, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(dx);
    (void)hipFree(dy);
    (void)hipModuleUnload(mod);
    return 0;
}
