// Gap between dependent kernels of one stream (development helper): 200 workgroups x 256 threads spinning ~40 us,
// start / end stamped with the 100 MHz wall clock; variants: alone, beside a stream of full-GPU kernels, with a
// write to mapped host memory at the end of every workgroup.
// hipcc --offload-arch=gfx950 -O2 tools/gap_probe.hip -o tools/gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_chain(long ticks, unsigned long long *clk, uint4 *host_rec, unsigned seq) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(&clk[0], t0);
    while (wall_clock64() - t0 < (unsigned long long)ticks) {}
    if (threadIdx.x == 0) {
        if (host_rec) host_rec[blockIdx.x] = make_uint4(1, 2, 3, seq);
        atomicMax(&clk[1], wall_clock64());
    }
}
__global__ void k_fill(float *p, long n, int reps) {  // full-GPU busy kernel with LDS, like the warp
    extern __shared__ float sm[];
    float acc = 0;
    for (int r = 0; r < reps; r++)
        for (long i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += p[i] * 1.0001f;
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (acc == 123.456f) p[0] = sm[0];
}
int main() {
    int lo, hi;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t t, w;
    (void)hipStreamCreateWithPriority(&t, hipStreamNonBlocking, hi);
    (void)hipStreamCreateWithFlags(&w, hipStreamNonBlocking);
    const int N = 300;
    unsigned long long *clk;
    (void)hipMalloc(&clk, sizeof(unsigned long long) * 2 * N);
    uint4 *hrec, *hrec_dev;
    (void)hipHostMalloc(&hrec, 16 * 256, hipHostMallocMapped);
    (void)hipHostGetDevicePointer((void **)&hrec_dev, hrec, 0);
    float *buf;
    const long nf = 8 << 20;
    (void)hipMalloc(&buf, nf * 4);
    (void)hipMemset(buf, 0, nf * 4);
    std::vector<unsigned long long> init(2 * N), out(2 * N);
    for (int variant = 0; variant < 4; variant++) {
        const bool busy = variant & 1, host_write = variant & 2;
        for (int i = 0; i < N; i++) init[2 * i] = ~0ull, init[2 * i + 1] = 0;
        (void)hipMemcpy(clk, init.data(), sizeof(unsigned long long) * 2 * N, hipMemcpyHostToDevice);
        (void)hipDeviceSynchronize();
        for (int i = 0; i < N; i++) {
            if (busy) hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 40 * 1024, w, buf, nf, 1);
            hipLaunchKernelGGL(k_chain, dim3(200), dim3(256), 0, t, 4000L, clk + 2 * i, host_write ? hrec_dev : nullptr, (unsigned)i);
        }
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(out.data(), clk, sizeof(unsigned long long) * 2 * N, hipMemcpyDeviceToHost);
        double dur = 0, gap = 0;
        for (int i = 50; i < N; i++) dur += (out[2 * i + 1] - out[2 * i]) * 0.01, gap += ((double)out[2 * i] - (double)out[2 * i - 1]) * 0.01;
        printf("busy-neighbour %d host-write %d: kernel %.1f us, gap to next %.1f us\n", (int)busy, (int)host_write, dur / (N - 50), gap / (N - 50));
    }
    return 0;
}
