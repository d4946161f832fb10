// vstab_device10.hpp -- device-side arithmetic of the 10-bit pixel path (BASELINE.json config 5), shared by the direct-gather
// kernel (vstab_warp_p010.hip) and the LDS-tiled one (vstab_warp_fused.hip).  No reference counterpart: the arithmetic is
// DEFINED by the test infrastructure's plain-C statement (its warp_p010 chain) and reproduced here bit for bit.
#pragma once
#include "vstab_device.hpp"
#include "vstab_internal.hpp"

namespace vstab {

__device__ __forceinline__ int sat10(int v) { return min(max(v, 0), 1023); }
// luma term + chroma term: the exact sum needs 33 bits (959 * CY + 511 * CUB = 2.25e9), but whenever it exceeds INT_MAX the
// pixel is saturated anyway ((2^31 - 1) >> 20 = 2047 > 1023), and it never goes below -1.09e9: a saturating 32-bit add
// (v_add_i32 ... clamp) gives the same channel value as the 64-bit sum of the definition.
__device__ __forceinline__ int channel10(int yy, int c) { return sat10(__builtin_elementwise_add_sat(yy, c) >> 20); }

// B, G, R in [0, 1023] of one tap from its luma sample and chroma pair (both still P010 words); !valid -> 0 (BORDER_CONSTANT)
__device__ __forceinline__ void convert_tap10(uint32_t ys, uint32_t c, bool valid, int &b, int &g, int &r) {
    const int yv = (int)(ys >> 6);
    const int u = (int)((c & 0xffffu) >> 6) - 512, v = (int)(c >> 22) - 512;
    const int yy = max(yv - 64, 0) * CY;  // <= 959 * 1220542 < 2^31
    b = valid ? channel10(yy, (1 << 19) + CUB * u) : 0;
    g = valid ? channel10(yy, (1 << 19) + CVG * v + CUG * u) : 0;
    r = valid ? channel10(yy, (1 << 19) + CVR * v) : 0;
}

struct Dword2 {  // 4-byte aligned pair of dwords / 2-byte aligned dword: the loads below are narrower than their natural alignment
    uint32_t lo, hi;
} __attribute__((packed, aligned(4)));
struct Dword1 {
    uint32_t v;
} __attribute__((packed, aligned(2)));

// The two horizontally adjacent taps (X, Yr), (X + 1, Yr) of a pixel's footprint with TWO loads: one dword holding both
// luma samples and one pair of dwords holding the (at most two) chroma pairs they use; positions are clamped into the
// row so that every load is in bounds, and taps outside the source come back as 0.  Needs sw >= 4.
template <typename Args>
__device__ __forceinline__ void fetch_row10(const Args &a, int X, int Yr, int &b0, int &g0, int &r0, int &b1, int &g1, int &r1) {
    const bool row_ok = (unsigned)Yr < (unsigned)a.sh;
    const int Yc = min(max(Yr, 0), a.sh - 1);
    const int col0 = min(max(X, 0), a.sw - 2);          // samples col0, col0 + 1
    const uint32_t yy = reinterpret_cast<const Dword1 *>(a.y + (size_t)Yc * a.pitch_y + 2 * (size_t)col0)->v;
    const int np = a.sw >> 1, pX = X >> 1, pX1 = (X + 1) >> 1;
    const int pc0 = min(max(pX, 0), np - 2);            // chroma pairs pc0, pc0 + 1
    const Dword2 cc = *reinterpret_cast<const Dword2 *>(a.uv + (size_t)(Yc >> 1) * a.pitch_uv + 4 * (size_t)pc0);
    const int d = X - col0;                             // 0 inside; -1 at X = -1; 1 at X = sw - 1
    const uint32_t yl = d == 1 ? yy >> 16 : yy & 0xffffu, yr = d == -1 ? yy & 0xffffu : yy >> 16;
    const uint32_t cl = pX - pc0 == 1 ? cc.hi : cc.lo, cr = pX1 - pc0 == 1 ? cc.hi : cc.lo;
    convert_tap10(yl << 0, cl, row_ok && (unsigned)X < (unsigned)a.sw, b0, g0, r0);
    convert_tap10(yr << 0, cr, row_ok && (unsigned)(X + 1) < (unsigned)a.sw, b1, g1, r1);
}

__device__ __forceinline__ int blend_fp16(int p00, int p01, int p10, int p11, int w00, int w01, int w10, int w11) {
    const _Float16 k = (_Float16)(1.0f / 1024.0f);
    _Float16 acc = (_Float16)0.0f;
    acc = __builtin_fmaf16((_Float16)p00, (_Float16)w00 * k, acc);  // (w * 2^-10 is exact: w <= 1024 has <= 11 significant bits)
    acc = __builtin_fmaf16((_Float16)p01, (_Float16)w01 * k, acc);
    acc = __builtin_fmaf16((_Float16)p10, (_Float16)w10 * k, acc);
    acc = __builtin_fmaf16((_Float16)p11, (_Float16)w11 * k, acc);
    return min((int)__builtin_rintf((float)acc), 1023);
}


// ---- the LDS-tiled kernel's pixel: B | G << 10 | R << 20 -------------------------------------------------------------
// chroma terms of one (U, V) pair of P010 words with the luma offset folded in: channel = sat10((max(y, 64) * CY + term) >> 20)
// is the same integer as sat10((max(y - 64, 0) * CY + (1 << 19) + C * uv) >> 20) -- also under the saturating add, whose
// result depends on the true sum only.
__device__ __forceinline__ ChromaTerm chroma_term10(uint32_t uv_words) {
    const int u = (int)((uv_words & 0xffffu) >> 6) - 512, v = (int)(uv_words >> 22) - 512;
    constexpr int K = (1 << 19) - 64 * CY;
    return {K + CVR * v, K + CVG * v + CUG * u, K + CUB * u};
}
// one pixel from its 10-bit luma value (sample >> 6) and folded chroma terms
__device__ __forceinline__ uint32_t pack_bgr10(int yv, const ChromaTerm &c) {
    const int t = __mul24(max(yv, 64), CY);  // <= 1023 * 1220542 < 2^31
    const int b = sat10(__builtin_elementwise_add_sat(t, c.buv) >> 20), g = sat10(__builtin_elementwise_add_sat(t, c.guv) >> 20),
              r = sat10(__builtin_elementwise_add_sat(t, c.ruv) >> 20);
    return (uint32_t)b | ((uint32_t)g << 10) | ((uint32_t)r << 20);
}
// the blend of four packed taps (both definitions); returns the packed result
template <int BLEND>
__device__ __forceinline__ uint32_t blend_bgr10(uint32_t t00, uint32_t t01, uint32_t t10, uint32_t t11, int fx, int fy) {
    const int gx = 32 - fx, gy = 32 - fy;
    const int w00 = gx * gy, w01 = fx * gy, w10 = gx * fy, w11 = fx * fy;
    uint32_t out = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int p00 = (int)((t00 >> (10 * c)) & 1023u), p01 = (int)((t01 >> (10 * c)) & 1023u), p10 = (int)((t10 >> (10 * c)) & 1023u),
                  p11 = (int)((t11 >> (10 * c)) & 1023u);
        int v;
        if constexpr (BLEND == VSTAB_BLEND_FP16) v = blend_fp16(p00, p01, p10, p11, w00, w01, w10, w11);
        else v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + 512) >> 10;
        out |= (uint32_t)v << (10 * c);
    }
    return out;
}
// ---- the LDS-tiled kernel's pixel for the fp16 blend: three binary16 values, B | G << 16 and R | 0 << 16 (8 bytes) ----------------
// The taps of VSTAB_BLEND_FP16 enter the fused multiply-adds as binary16 numbers.  Converting them there costs twelve int -> half
// conversions (and twelve field extracts) per OUTPUT pixel; staged as halves they are converted once per SOURCE pixel, and the
// blend is eight packed fused multiply-adds (B and G together, R beside a zero lane).  0 .. 1023 are exact in binary16.
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 pack_bgr10h(int yv, const ChromaTerm &c) {
    const int t = __mul24(max(yv, 64), CY);
    const int b = sat10(__builtin_elementwise_add_sat(t, c.buv) >> 20), g = sat10(__builtin_elementwise_add_sat(t, c.guv) >> 20),
              r = sat10(__builtin_elementwise_add_sat(t, c.ruv) >> 20);
    const half2v bg = {(_Float16)(unsigned short)b, (_Float16)(unsigned short)g}, r0 = {(_Float16)(unsigned short)r, (_Float16)0.0f};
    return make_uint2(__builtin_bit_cast(uint32_t, bg), __builtin_bit_cast(uint32_t, r0));
}
// the definition's four fused multiply-adds per channel (taps 00, 01, 10, 11; weights w / 1024, exact in binary16), packed; then
// round to nearest even and clamp: min(acc, 1023) + 1024 has ulp 1, so the adder does the rounding and the integer is the low ten
// bits of the sum's pattern (0x6400 + n).  Returns B | G << 10 | R << 20 like blend_bgr10.
__device__ __forceinline__ uint32_t blend_bgr10h(uint2 t00, uint2 t01, uint2 t10, uint2 t11, int fx, int fy) {
    // the weights w / 1024 = (gx / 32) (gy / 32) etc.: the factors are multiples of 1/32 up to 1 and every product a multiple of 1/1024 up
    // to 1 -- all exact in binary16 -- so two packed multiplies give the four weights the definition writes as (_Float16)w * 2^-10
    const _Float16 k = (_Float16)(1.0f / 32.0f);
    const _Float16 fxh = (_Float16)(short)fx * k, fyh = (_Float16)(short)fy * k;
    const half2v wx = {(_Float16)1.0f - fxh, fxh};                                    // (gx, fx) / 32
    const half2v wt = wx * (half2v){(_Float16)1.0f - fyh, (_Float16)1.0f - fyh};      // (w00, w01)
    const half2v wb = wx * (half2v){fyh, fyh};                                        // (w10, w11)
    const half2v w00 = {wt.x, wt.x}, w01 = {wt.y, wt.y}, w10 = {wb.x, wb.x}, w11 = {wb.y, wb.y}, zero = {(_Float16)0.0f, (_Float16)0.0f};
    auto h2 = [](uint32_t v) { return __builtin_bit_cast(half2v, v); };
    half2v bg = __builtin_elementwise_fma(h2(t00.x), w00, zero), r = __builtin_elementwise_fma(h2(t00.y), w00, zero);
    bg = __builtin_elementwise_fma(h2(t01.x), w01, bg), r = __builtin_elementwise_fma(h2(t01.y), w01, r);
    bg = __builtin_elementwise_fma(h2(t10.x), w10, bg), r = __builtin_elementwise_fma(h2(t10.y), w10, r);
    bg = __builtin_elementwise_fma(h2(t11.x), w11, bg), r = __builtin_elementwise_fma(h2(t11.y), w11, r);
    const half2v top = {(_Float16)1023.0f, (_Float16)1023.0f}, magic = {(_Float16)1024.0f, (_Float16)1024.0f};
    const uint32_t bgi = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(bg, top) + magic) & 0x03ff03ffu;
    const uint32_t ri = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(r, top) + magic) & 0x000003ffu;
    return (bgi & 0x3ffu) | ((bgi >> 6) & 0xffc00u) | (ri << 20);
}

// One output pixel straight from global memory (the rare path of the tiled kernel): the direct kernel's arithmetic.
template <int BLEND, typename Args>
__device__ __forceinline__ uint32_t gather_pixel10(const Args &a, int sx, int sy) {
    const int X = sx >> 5, Y = sy >> 5, fx = sx & 31, fy = sy & 31;
    if (X >= a.sw || X + 1 < 0 || Y >= a.sh || Y + 1 < 0) return 0;
    const int w00 = (32 - fx) * (32 - fy), w01 = fx * (32 - fy), w10 = (32 - fx) * fy, w11 = fx * fy;
    int b0, g0, r0, b1, g1, r1, b2, g2, r2, b3, g3, r3;
    fetch_row10(a, X, Y, b0, g0, r0, b1, g1, r1);
    fetch_row10(a, X, Y + 1, b2, g2, r2, b3, g3, r3);
    int B, G, R;
    if constexpr (BLEND == VSTAB_BLEND_FP16) {
        B = blend_fp16(b0, b1, b2, b3, w00, w01, w10, w11), G = blend_fp16(g0, g1, g2, g3, w00, w01, w10, w11), R = blend_fp16(r0, r1, r2, r3, w00, w01, w10, w11);
    } else {
        B = (b0 * w00 + b1 * w01 + b2 * w10 + b3 * w11 + 512) >> 10, G = (g0 * w00 + g1 * w01 + g2 * w10 + g3 * w11 + 512) >> 10,
        R = (r0 * w00 + r1 * w01 + r2 * w10 + r3 * w11 + 512) >> 10;
    }
    return (uint32_t)B | ((uint32_t)G << 10) | ((uint32_t)R << 20);
}

}  // namespace vstab
