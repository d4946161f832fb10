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
#include <mutex>
#include <string>

namespace {

// The code object is loaded ONCE per process and path and never unloaded.  A hipModuleLoad / hipModuleUnload pair per call was what this
// file did until round 5: on this ROCm a module loaded lazily LATER (the product's own fat binary, at its first kernel launch) can then be
// copied to a device address the unloaded code object had occupied, which the GPU still maps read-only -- "Memory access fault ... Write
// access to a read-only page" during that load, in one run out of three or four (24 load / unload cycles of the reference kernel, then the
// first launch of any kernel of libvstab.so: no kernel of the library need have run before).
hipError_t reference_kernel(const char *co_path, hipFunction_t *fn) {
    static std::mutex m;
    static std::string loaded_path;
    static hipModule_t mod = nullptr;
    static hipFunction_t cached = nullptr;
    std::lock_guard<std::mutex> lk(m);
    if (!cached || loaded_path != co_path) {
        hipModule_t fresh = nullptr;  // (another path: a second module; the first one stays loaded)
        hipError_t e = hipModuleLoad(&fresh, co_path);
        if (e != hipSuccess) return e;
        hipFunction_t f = nullptr;
        e = hipModuleGetFunction(&f, fresh, "createMap");
        if (e != hipSuccess) return e;
        mod = fresh, cached = f, loaded_path = co_path;
    }
    *fn = cached;
    return hipSuccess;
}

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
    void *dx = nullptr, *dy = nullptr;
    if (cols <= 0 || rows <= 0 || cols > 32767 || rows > 32767 || block_x <= 0 || block_y <= 0 || block_x * block_y > 1024) {
        if (err && errlen > 0) snprintf(err, (size_t)errlen, "bad size");
        return -1;
    }
    hipFunction_t fn = nullptr;
    RC_TRY(reference_kernel(co_path, &fn));
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
    return 0;
}

// The rolling-shutter variant's checker (config 5 has no reference counterpart; the definition -- include/vstab.h, vstab_warp_nv12_rs
// -- gives every output row its own matrix m(y) and says the row is "what the reference's kernel returns for that row when it is
// handed m(y) as its rotation").  Literally that: for every row y the reference kernel is launched over the whole frame with
// m(y) in its nine rotation arguments and row y of its output is kept.  m_k(y) = fmaf((float)y / (float)max(rows - 1, 1),
// rot_bottom[k] - params[8 + k], params[8 + k]) in IEEE binary32 (compiled with -ffp-contract=off; SSE division and glibc fmaf).
#include <cmath>
extern "C" __attribute__((visibility("default"))) int refcl_create_map_rs(const char *co_path, int cols, int rows, const float params[17],
                                                                          const float rot_bottom[9], float *out_x, float *out_y, char *err,
                                                                          int errlen) {
    void *dx = nullptr, *dy = nullptr;  // scratch planes of one launch
    void *rx = nullptr, *ry = nullptr;  // the assembled result
#undef RC_TRY
#define RC_TRY(expr)                                          \
    do {                                                      \
        hipError_t e_ = (expr);                               \
        if (e_ != hipSuccess) {                               \
            for (void *p_ : {dx, dy, rx, ry})                 \
                if (p_) (void)hipFree(p_);                    \
            return fail(err, errlen, #expr, e_);              \
        }                                                     \
    } while (0)
    if (cols <= 0 || rows <= 0 || cols > 32767 || rows > 32767) {
        if (err && errlen > 0) snprintf(err, (size_t)errlen, "bad size");
        return -1;
    }
    hipFunction_t fn = nullptr;
    RC_TRY(reference_kernel(co_path, &fn));
    const size_t row_bytes = (size_t)cols * sizeof(float), bytes = row_bytes * rows;
    RC_TRY(hipMalloc(&dx, bytes));
    RC_TRY(hipMalloc(&dy, bytes));
    RC_TRY(hipMalloc(&rx, bytes));
    RC_TRY(hipMalloc(&ry, bytes));
    RC_TRY(hipMemset(rx, 0xff, bytes));
    RC_TRY(hipMemset(ry, 0xff, bytes));
    float d[9];
    for (int k = 0; k < 9; k++) d[k] = rot_bottom[k] - params[8 + k];
    const float den = (float)(rows > 1 ? rows - 1 : 1);
    const unsigned gx = (unsigned)((cols + 63) / 64), gy = (unsigned)((rows + 3) / 4);
    for (int y = 0; y < rows; y++) {
        CreateMapArgs a;
        a.map_x = dx, a.step_x = cols * 4, a.offset_x = 0, a.rows = rows, a.cols = cols;
        a.map_y = dy, a.step_y = cols * 4, a.offset_y = 0;
        memcpy(a.f, params, sizeof(a.f));
        const float t = (float)y / den;
        for (int k = 0; k < 9; k++) a.f[8 + k] = std::fmaf(t, d[k], params[8 + k]);
        size_t size = sizeof(a);
        void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
        RC_TRY(hipModuleLaunchKernel(fn, gx, gy, 1, 64, 4, 1, 0, nullptr, nullptr, extra));  // (the kernarg buffer is copied at launch)
        RC_TRY(hipMemcpyAsync(static_cast<char *>(rx) + (size_t)y * row_bytes, static_cast<char *>(dx) + (size_t)y * row_bytes, row_bytes, hipMemcpyDeviceToDevice, nullptr));
        RC_TRY(hipMemcpyAsync(static_cast<char *>(ry) + (size_t)y * row_bytes, static_cast<char *>(dy) + (size_t)y * row_bytes, row_bytes, hipMemcpyDeviceToDevice, nullptr));
    }
    RC_TRY(hipDeviceSynchronize());
    RC_TRY(hipMemcpy(out_x, rx, bytes, hipMemcpyDeviceToHost));
    RC_TRY(hipMemcpy(out_y, ry, bytes, hipMemcpyDeviceToHost));
    for (void *p_ : {dx, dy, rx, ry}) (void)hipFree(p_);
    return 0;
}
