// Store-pattern microbenchmark (development): how fast can a 3524 x 1999 BGR8 frame (pitch 10572) be WRITTEN in the tile
// order of k_warp_fused, by store shape?  Build: hipcc --offload-arch=gfx950 -O3 -o probe_store probe_store.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int DW = 3524, DH = 1999, PITCH = DW * 3;

// V = 0: one dword per lane, 48 lanes per 64-pixel row (the kernel's current store)
// V = 1: dwordx4 per lane: a wave's 8 rows x 192 B = 96 chunks of 16 B, two instructions
// V = 2: dwordx2 per lane: 192 chunks of 8 B, three instructions
template <int V>
__global__ void __launch_bounds__(256) k_store(uint8_t *dst, int tiles_x, int tiles_y, uint32_t seed) {
    const int tile = blockIdx.x, ty = tile / tiles_x, tx = tile - ty * tiles_x;
    if (ty >= tiles_y) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = tx * 64, y0 = ty * 32 + wave * 8;
    const int ncols = min(64, DW - x0), nbytes = 3 * ncols;
    const uint32_t v = seed * 2654435761u + threadIdx.x + tile;
    if (V == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int y = y0 + j;
            if (y >= DH) break;
            uint8_t *o = dst + (size_t)y * PITCH + (size_t)x0 * 3;
            if (4 * lane + 4 <= nbytes) *reinterpret_cast<uint32_t *>(o + 4 * lane) = v + j;
        }
    } else if (V == 1) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int c = i * 64 + lane;  // chunk of 16 B: row c / 12, byte offset (c % 12) * 16
            const int r = c / 12, k = c - r * 12, y = y0 + r;
            if (c < 96 && y < DH && 16 * k + 16 <= nbytes) {
                uint8_t *o = dst + (size_t)y * PITCH + (size_t)x0 * 3 + 16 * k;
                *reinterpret_cast<uint4 *>(o) = make_uint4(v, v + 1, v + 2, v + i);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int c = i * 64 + lane;  // chunk of 8 B: row c / 24
            const int r = c / 24, k = c - r * 24, y = y0 + r;
            if (y < DH && 8 * k + 8 <= nbytes) {
                uint8_t *o = dst + (size_t)y * PITCH + (size_t)x0 * 3 + 8 * k;
                *reinterpret_cast<uint2 *>(o) = make_uint2(v, v + i);
            }
        }
    }
}

int main() {
    const int NB = 16;
    std::vector<uint8_t *> bufs(NB);
    for (auto &b : bufs) CK(hipMalloc(&b, (size_t)PITCH * DH + 64));
    const int tiles_x = (DW + 63) / 64, tiles_y = (DH + 31) / 32;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int v = 0; v < 3; v++) {
        for (int rep = 0; rep < 2; rep++) {
            const int n = 100;
            CK(hipEventRecord(e0));
            for (int i = 0; i < n; i++) {
                uint8_t *d = bufs[i % NB];
                if (v == 0) hipLaunchKernelGGL(k_store<0>, dim3(tiles_x * tiles_y), dim3(256), 0, 0, d, tiles_x, tiles_y, (uint32_t)i);
                if (v == 1) hipLaunchKernelGGL(k_store<1>, dim3(tiles_x * tiles_y), dim3(256), 0, 0, d, tiles_x, tiles_y, (uint32_t)i);
                if (v == 2) hipLaunchKernelGGL(k_store<2>, dim3(tiles_x * tiles_y), dim3(256), 0, 0, d, tiles_x, tiles_y, (uint32_t)i);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("store shape %d (%s): %.2f us per frame, %.0f GB/s\n", v, v == 0 ? "dword x 48 lanes per row" : v == 1 ? "dwordx4, 2 per 8 rows" : "dwordx2, 3 per 8 rows",
                            ms * 1000 / n, (double)PITCH * DH / (ms / n * 1e-3) / 1e9);
        }
    }
    return 0;
}
