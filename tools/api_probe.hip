// HIP API cost probe (development helper): host time and sustained rate of launches / event operations.
// hipcc --offload-arch=gfx950 -O2 tools/api_probe.hip -o tools/api_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_empty(int *p) { if (p && threadIdx.x == 1000) *p = 1; }
__global__ void k_spin(long cycles) { const long t0 = clock64(); while (clock64() - t0 < cycles) {} }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t a, b, c;
    int lo, hi;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&a, hipStreamNonBlocking, hi);
    hipStreamCreateWithPriority(&b, hipStreamNonBlocking, lo);
    hipStreamCreateWithFlags(&c, hipStreamNonBlocking);
    const int N = 2000;
    std::vector<hipEvent_t> ev(64);
    for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, a, nullptr);
    hipDeviceSynchronize();
    // 1. empty launches, one stream
    double t0 = now();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, a, nullptr);
    double t1 = now();
    hipDeviceSynchronize();
    double t2 = now();
    printf("1 stream, empty kernel:           host %.2f us/launch, sustained %.2f us/launch\n", (t1 - t0) / N, (t2 - t0) / N);
    // 2. launches alternating over three streams, no dependencies
    t0 = now();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, i % 3 == 0 ? a : i % 3 == 1 ? b : c, nullptr);
    t1 = now();
    hipDeviceSynchronize();
    t2 = now();
    printf("3 streams, independent:           host %.2f us/launch, sustained %.2f us/launch\n", (t1 - t0) / N, (t2 - t0) / N);
    // 3. launch + event record on the same stream
    t0 = now();
    for (int i = 0; i < N; i++) {
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, a, nullptr);
        hipEventRecord(ev[i % 64], a);
    }
    t1 = now();
    hipDeviceSynchronize();
    t2 = now();
    printf("launch + eventRecord:             host %.2f us/iter, sustained %.2f us/iter\n", (t1 - t0) / N, (t2 - t0) / N);
    // 4. producer on b -> event -> consumer on c waits (the ingest -> warp pattern), 10 us kernels
    for (long cyc : {0L, 1000L, 20000L}) {
        t0 = now();
        for (int i = 0; i < N; i++) {
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, cyc);
            hipEventRecord(ev[i % 64], b);
            hipStreamWaitEvent(c, ev[i % 64], 0);
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, c, cyc);
        }
        t1 = now();
        hipDeviceSynchronize();
        t2 = now();
        printf("b:kernel,record; c:wait,kernel (%5ld cyc spin): host %.2f us/iter, sustained %.2f us/iter\n", cyc, (t1 - t0) / N, (t2 - t0) / N);
    }
    // 5. same plus the reverse dependency (slot reuse: b waits for c's event of 8 iterations ago)
    std::vector<hipEvent_t> ev2(64);
    for (auto &e : ev2) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    t0 = now();
    for (int i = 0; i < N; i++) {
        if (i >= 8) hipStreamWaitEvent(b, ev2[(i - 8) % 64], 0);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, 1000L);
        hipEventRecord(ev[i % 64], b);
        hipStreamWaitEvent(c, ev[i % 64], 0);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, c, 1000L);
        hipEventRecord(ev2[i % 64], c);
    }
    t1 = now();
    hipDeviceSynchronize();
    t2 = now();
    printf("with reverse dependency 8 back:   host %.2f us/iter, sustained %.2f us/iter\n", (t1 - t0) / N, (t2 - t0) / N);
    // 6. hipEventQuery / hipEventSynchronize on a completed event
    t0 = now();
    for (int i = 0; i < N; i++) (void)hipEventQuery(ev[0]);
    t1 = now();
    for (int i = 0; i < N; i++) (void)hipEventSynchronize(ev[0]);
    t2 = now();
    printf("completed event: query %.2f us, synchronize %.2f us\n", (t1 - t0) / N, (t2 - t1) / N);
    return 0;
}
