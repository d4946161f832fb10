// Development probe: semantics of a few gfx950 instructions used by the warp kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <cstring>

__global__ void k_ashr_pk(const int *a, const int *b, uint32_t *out, uint32_t prefill) {
    int i = threadIdx.x;
    uint32_t d = prefill, d2 = prefill;
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 20" : "+v"(d) : "v"(a[i]), "v"(b[i]));
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 20 op_sel:[0,0,0,1]" : "+v"(d2) : "v"(a[i]), "v"(b[i]));
    out[2 * i] = d;
    out[2 * i + 1] = d2;
}

// hand-rolled IEEE division / sqrt / reciprocal vs compiler's correctly rounded ones
__device__ __forceinline__ float div_fast(float n, float d) {
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = n * r;
    float e1 = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e1, r, q);
    float e2 = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e2, r, q);
}
__device__ __forceinline__ float div_fast1(float n, float d) {  // one quotient refinement only
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = n * r;
    float e1 = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e1, r, q);
}
__device__ __forceinline__ float sqrt_fast(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    float sd = __int_as_float(__float_as_int(s) - 1), su = __int_as_float(__float_as_int(s) + 1);
    float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
    s = rd <= 0.0f ? sd : s;
    s = ru > 0.0f ? su : s;
    return s;
}
// rsq-based correctly rounded sqrt (LLVM's expansion for the flush-denormal mode)
__device__ __forceinline__ float sqrt_rsq(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    float e = __builtin_fmaf(-h, g, 0.5f);
    h = __builtin_fmaf(h, e, h);
    g = __builtin_fmaf(g, e, g);
    float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__global__ void k_div(const float *n, const float *d, int count, unsigned long long *bad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float ref = n[i] / d[i];
    float a = div_fast(n[i], d[i]), b = div_fast1(n[i], d[i]);
    if (__float_as_uint(a) != __float_as_uint(ref) && !(a != a && ref != ref)) atomicAdd(&bad[0], 1ull);
    if (__float_as_uint(b) != __float_as_uint(ref) && !(b != b && ref != ref)) atomicAdd(&bad[1], 1ull);
    float x = fabsf(n[i]);
    float sr = sqrtf(x), sf = sqrt_fast(x);
    if (__float_as_uint(sr) != __float_as_uint(sf)) atomicAdd(&bad[2], 1ull);
    float sq = sqrt_rsq(x);
    if (__float_as_uint(sr) != __float_as_uint(sq) && x != 0.0f) atomicAdd(&bad[4], 1ull);
    float rr = 1.0f / d[i], rf = div_fast(1.0f, d[i]);
    if (__float_as_uint(rr) != __float_as_uint(rf) && !(rr != rr && rf != rf)) atomicAdd(&bad[3], 1ull);
}

int main() {
    const int N = 64;
    int ha[N], hb[N];
    for (int i = 0; i < N; i++) {
        ha[i] = (i - 20) * (17 << 20) + 12345;  // >>20 gives (i-20)*17: negative .. > 255
        hb[i] = (40 - i) * (9 << 20) + 777;
    }
    int *a, *b; uint32_t *o;
    hipMalloc(&a, sizeof(ha)); hipMalloc(&b, sizeof(hb)); hipMalloc(&o, N * 8);
    hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
    uint32_t ho[2 * N];
    for (uint32_t prefill : {0u, 0xA5A5A5A5u}) {
        k_ashr_pk<<<1, N>>>(a, b, o, prefill);
        hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        printf("prefill %08x\n", prefill);
        for (int i = 0; i < N; i += 5)
            printf("  a>>20=%5d b>>20=%5d  lo-form=%08x  hi-form=%08x\n", ha[i] >> 20, hb[i] >> 20, ho[2 * i], ho[2 * i + 1]);
    }
    // division / sqrt exactness on the warp kernel's domain
    const int M = 1 << 24;
    float *hn = (float *)malloc(M * 4), *hd = (float *)malloc(M * 4);
    srand(1);
    for (int i = 0; i < M; i++) {
        float u = (float)rand() / RAND_MAX, v = (float)rand() / RAND_MAX;
        int mode = i & 3;
        hn[i] = mode == 0 ? (u * 8 - 4) : mode == 1 ? (u - 0.5f) * 1e-3f : mode == 2 ? u * 20 : (u - 0.5f) * 200;
        hd[i] = mode == 0 ? (v * 2 + 0.05f) : mode == 1 ? v + 0.1f : mode == 2 ? (v * 6 + 1e-3f) : (v - 0.5f) * 3;
    }
    float *dn, *dd; unsigned long long *bad, hbad[5];
    hipMalloc(&dn, M * 4); hipMalloc(&dd, M * 4); hipMalloc(&bad, 40); hipMemset(bad, 0, 40);
    hipMemcpy(dn, hn, M * 4, hipMemcpyHostToDevice); hipMemcpy(dd, hd, M * 4, hipMemcpyHostToDevice);
    k_div<<<M / 256, 256>>>(dn, dd, M, bad);
    hipMemcpy(hbad, bad, 40, hipMemcpyDeviceToHost);
    printf("of %d: div 2-step mismatches %llu, div 1-step mismatches %llu, sqrt mismatches %llu, rcp mismatches %llu, rsq-sqrt mismatches %llu\n", M, hbad[0], hbad[1], hbad[2], hbad[3], hbad[4]);
    return 0;
}
