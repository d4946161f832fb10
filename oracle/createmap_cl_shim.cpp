// createmap_cl_shim.cpp -- host driver for the reference's OWN kernel text (TEST INFRASTRUCTURE).
//
// oracle/Makefile compiles /root/reference/opencv/createMap.cl, unmodified and in place, to an
// x86-64 object with ROCm clang (-x cl).  That object leaves exactly five OpenCL built-ins
// undefined; this file supplies them with their OpenCL 1.2 meaning (s6.12.1, 6.12.2, 6.12.3,
// 6.12.5) in plain IEEE binary32 and a loop that plays the NDRange
// (FrameSourceWarp.cpp:278 global = {cols, rows}).  The result, oracle/_ref/libcreatemap_ref.so,
// validates oracle/vstab_oracle.c:vo_create_map and generated tests/golden/createmap_*.npz.
// It never travels as source: only the built .so (git-ignored) and the golden vectors do.
#include <cmath>
#include <cstddef>

typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float3 __attribute__((ext_vector_type(3)));

static thread_local size_t g_gid[3];

size_t get_global_id(unsigned int d) { return d < 3 ? g_gid[d] : 0; }
float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
float atan(float x) { return atanf(x); }
int mad24(int a, int b, int c) { return a * b + c; }
float length(float2 v) { return sqrtf(v.x * v.x + v.y * v.y); }

// createMap.cl:1-9 signature (address-space pointers are plain pointers on x86-64)
extern "C" void createMap(float *out_map_x, int map_x_step, int map_x_offset, int map_rows,
                          int map_cols, float *out_map_y, int map_y_step, int map_y_offset,
                          float src_center_x, float src_center_y, float src_focal_x,
                          float src_focal_y, float map_center_x, float map_center_y,
                          float map_focal_x, float map_focal_y, float rot00, float rot01,
                          float rot02, float rot10, float rot11, float rot12, float rot20,
                          float rot21, float rot22);

// p[17] in kernel-argument order (FrameSourceWarp.cpp:283-299); dense cols-wide planes.
extern "C" __attribute__((visibility("default"))) void createmap_ref_run(float *mapx, float *mapy,
                                                                        int cols, int rows,
                                                                        const float *p) {
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            g_gid[0] = (size_t)x, g_gid[1] = (size_t)y;
            createMap(mapx, cols * 4, 0, rows, cols, mapy, cols * 4, 0, p[0], p[1], p[2], p[3], p[4],
                      p[5], p[6], p[7], p[8], p[9], p[10], p[11], p[12], p[13], p[14], p[15], p[16]);
        }
}
