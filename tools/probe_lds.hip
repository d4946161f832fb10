// Development probe: LDS instruction cost by access width and ALIGNMENT on gfx950 (cycles per wave-instruction and CU).
// The plane-wise warp reads taps at byte granularity; this measures what an unaligned ds_read_u16 / ds_read_b32 costs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define REP8(X) X X X X X X X X
template <int OP>
__global__ void __launch_bounds__(256) k_lds(unsigned long long *stamp, int iters, int mul, int add) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds + 1024 + lane * mul + add;
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#define BODY(INS) REP8(asm volatile(INS " %0, %8\n" INS " %1, %8 offset:256\n" INS " %2, %8 offset:512\n" INS " %3, %8 offset:768\n" INS " %4, %8 offset:1024\n" INS " %5, %8 offset:1280\n" INS " %6, %8 offset:1536\n" INS " %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr));)
#define BODYW(INS) REP8(asm volatile(INS " %8, %0\n" INS " %8, %1 offset:256\n" INS " %8, %2 offset:512\n" INS " %8, %3 offset:768\n" INS " %8, %4 offset:1024\n" INS " %8, %5 offset:1280\n" INS " %8, %6 offset:1536\n" INS " %8, %7 offset:1792\n s_waitcnt lgkmcnt(0)" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(addr) : "memory");)
        if (OP == 0) { BODY("ds_read_u8") }
        if (OP == 1) { BODY("ds_read_u16") }
        if (OP == 2) { BODY("ds_read_b32") }
        if (OP == 3) { BODYW("ds_write_b8") }
        if (OP == 4) { BODYW("ds_write_b16") }
        if (OP == 5) { BODYW("ds_write_b32") }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[0] = t1 - t0, stamp[1] = r1 - r0;
    if (iters < 0) stamp[2] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
typedef void (*kern_t)(unsigned long long *, int, int, int);
int main() {
    unsigned long long *stamp; hipMalloc(&stamp, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char *name; kern_t k; } ops[] = {{"ds_read_u8", k_lds<0>}, {"ds_read_u16", k_lds<1>}, {"ds_read_b32", k_lds<2>}, {"ds_write_b8", k_lds<3>}, {"ds_write_b16", k_lds<4>}, {"ds_write_b32", k_lds<5>}};
    struct { const char *name; int mul, add; } pats[] = {{"lane*4", 4, 0}, {"lane*4+1", 4, 1}, {"lane*4+2", 4, 2}, {"lane*4+3", 4, 3}, {"lane*2", 2, 0}, {"lane*2+1", 2, 1}, {"lane*1", 1, 0}, {"lane*1+1", 1, 1}, {"lane*3", 3, 0}, {"lane*8", 8, 0}, {"same", 0, 0}, {"same+1", 0, 1}};
    const int W = 4, iters = 200;   // 4 workgroups of 4 waves per CU: the LDS pipe of a CU is shared by its four SIMDs
    printf("# cycles of the CU's LDS pipe per wave-instruction = launch duration x shader clock / (64 instr x iters x 4 waves x %d workgroups per CU)\n", W);
    printf("%-14s", "# address:");
    for (auto &p : pats) printf(" %9s", p.name);
    printf("\n");
    for (auto &o : ops) {
        printf("%-14s", o.name);
        for (auto &p : pats) {
            hipFuncSetAttribute((const void *)o.k, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
            hipLaunchKernelGGL(o.k, dim3(256 * W), dim3(256), 32768, 0, stamp, 10, p.mul, p.add);
            hipEventRecord(e0);
            hipLaunchKernelGGL(o.k, dim3(256 * W), dim3(256), 32768, 0, stamp, iters, p.mul, p.add);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, stamp, 16, hipMemcpyDeviceToHost);
            const double mhz = (double)h[0] / ((double)h[1] / 100.0);
            printf(" %9.2f", (double)ms * 1e-3 * mhz * 1e6 / (64.0 * iters * 4 * W));
        }
        printf("\n");
    }
    return 0;
}
