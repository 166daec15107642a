// pow(x, p) for the transfer stage (TransferFunction.java:39-44, 83-87: Math.pow on doubles, result cast to float).
// The reference's Math.pow is a HotSpot intrinsic specified to 1 ulp of the DOUBLE result; what reaches the pixels is
// its float cast, so any double result within ~1e-13 relative of the true power gives the same float except when the
// true value lies within 1e-13 of a float rounding boundary (probability ~2e-6 per sample; the stage's stated tolerance
// is 1 float ulp, tests/test_stages_gpu.py). ocml's pow() carries extended precision through log and exp to be
// correctly rounded in double: 463 instructions (309 f64) for the PQ curve against ~110 here.
//   x = m 2^e, m in [sqrt(1/2), sqrt(2));  log2 m = (2/ln2) atanh(t), t = (m-1)/(m+1), |t| <= 0.1716: odd series to t^21
//   2^z = 2^n 2^r, n = rint(z), |r| <= 1/2: Taylor series of exp(r ln2) to r^13
// Measured against 80-bit long double on 6e5 inputs (numpy restatement with this header's coefficients, tests/test_fastpow_cpu.py): max relative error
// 2e-15 (p = 0.159), 9e-14 (p = 78.84 over the full float range; 1e-15 on PQ's actual base range [0.83, 1.01]);
// the float results of the whole PQ and sRGB curves were identical to libm's for all 2.2e6 sampled inputs.
#pragma once
#include <hip/hip_runtime.h>

namespace jxl {

// 1 / b to full double precision (v_rcp_f64 + two Newton steps); b finite, non-zero, normal
__device__ __forceinline__ double fp_rcp(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    return r;
}

// a / b within 1 ulp (quotient + one residual correction)
__device__ __forceinline__ double fp_div(double a, double b) {
    const double r = fp_rcp(b);
    const double q = a * r;
    return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}

// x finite and > 0
__device__ __forceinline__ double fp_pow_pos(double x, double p) {
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    if (m < 0.70710678118654757) {
        m = m * 2.0;
        e = e - 1;
    }
    const double num = m - 1.0, den = m + 1.0;
    const double r = fp_rcp(den);
    double t = num * r;
    t = __builtin_fma(__builtin_fma(-den, t, num), r, t);
    const double t2 = t * t;
    double P = 0.13739952770371081;  // (2 / ln 2) / (2k + 1), k = 10 .. 0
    P = __builtin_fma(P, t2, 0.15186263588304877);
    P = __builtin_fma(P, t2, 0.16972882833987804);
    P = __builtin_fma(P, t2, 0.19235933878519512);
    P = __builtin_fma(P, t2, 0.2219530832136867);
    P = __builtin_fma(P, t2, 0.26230818925253879);
    P = __builtin_fma(P, t2, 0.3205988979753252);
    P = __builtin_fma(P, t2, 0.41219858311113244);
    P = __builtin_fma(P, t2, 0.57707801635558531);
    P = __builtin_fma(P, t2, 0.96179669392597567);
    P = __builtin_fma(P, t2, 2.8853900817779268);
    const double L = __builtin_fma(t, P, (double)e);  // log2 x
    const double z = p * L;
    const double n = __builtin_rint(z);
    const double rr = z - n;
    double Q = 1.3691488853904124e-12;  // (ln 2)^k / k!, k = 13 .. 0
    Q = __builtin_fma(Q, rr, 2.5678435993488196e-11);
    Q = __builtin_fma(Q, rr, 4.4455382718708101e-10);
    Q = __builtin_fma(Q, rr, 7.0549116208011209e-09);
    Q = __builtin_fma(Q, rr, 1.0178086009239696e-07);
    Q = __builtin_fma(Q, rr, 1.3215486790144305e-06);
    Q = __builtin_fma(Q, rr, 1.5252733804059838e-05);
    Q = __builtin_fma(Q, rr, 0.00015403530393381606);
    Q = __builtin_fma(Q, rr, 0.0013333558146428441);
    Q = __builtin_fma(Q, rr, 0.0096181291076284769);
    Q = __builtin_fma(Q, rr, 0.055504108664821576);
    Q = __builtin_fma(Q, rr, 0.24022650695910069);
    Q = __builtin_fma(Q, rr, 0.69314718055994529);
    Q = __builtin_fma(Q, rr, 1.0);
    // |z| is at most a few thousand here; clamp so that the int conversion is defined, ldexp saturates to 0 / inf
    const int ni = (int)__builtin_fmin(__builtin_fmax(n, -4000.0), 4000.0);
    return __builtin_amdgcn_ldexp(Q, ni);
}

// Math.pow(x, p) for a positive non-integer p: +0 for x = +-0, +inf for x = +-inf, NaN for negative finite x and NaN
__device__ __forceinline__ double fp_pow(double x, double p) {
    if (x > 0.0 && x < __builtin_inf()) return fp_pow_pos(x, p);
    if (x == 0.0) return 0.0;
    if (x == __builtin_inf() || x == -__builtin_inf()) return __builtin_inf();
    return __builtin_nan("");
}

// TF_PQ.fromLinear (TransferFunction.java:83-87), with the reference's constants
__device__ __forceinline__ float fp_tf_pq(float f) {
    const double d = fp_pow((double)f, 0.159423828125);
    const double a = 0.8359375 + 18.8515625 * d, b = 1.0 + 18.6875 * d;
    // d = +inf gives inf / inf = NaN in the reference; fp_div would too (rcp(inf) = 0, inf * 0 = NaN)
    return (float)fp_pow(fp_div(a, b), 78.84375);
}

// ---- PQ through a table: one 16-byte gather and three fused multiply-adds per sample ----------------------------------------
// The double-precision form above costs ~110 f64 operations per sample (f64 issues at half the f32 rate): three samples per
// pixel made the transfer stage of the 8K PQ configuration dearer than Gaborish + two EPF iterations + XYB together
// (round-1 verdict). PQ is ONE fixed smooth function float -> float, so it is tabulated: the positive floats of [2^-40, 4) are
// cut into 42 binades x 128 mantissa segments; per segment the host fits y(xm + t) = a0 + a1 t + a2 t^2 in long double through
// the three Chebyshev nodes of the segment (truncation error = third derivative x h^3 / 24: 0.013 ulp; 64 segments per binade gave 0.1) and stores
// a0 as a float pair (hi, lo), a1, a2 as floats. t = x - xm is exact (both in one binade), the correction a0lo + t (a1 + t a2)
// is below 0.3 % of a0hi, so its float rounding is ~2^-31 relative and the only rounding that matters is the final addition:
// the result is within 0.5 + 0.01 ulp of the true value, the reference's (float) of a double is within 0.5 ulp -> they differ
// by at most 1 ulp (checked over ALL 2^32 float inputs on the GPU against the oracle: tools/pq_sweep.py, profiles/). Inputs
// outside the table (zero, tiny, >= 4, negative, inf, NaN) take the double-precision form.
constexpr int kPqExpLo = 87;   // biased exponent of 2^-40
constexpr int kPqExpHi = 129;  // first biased exponent beyond the table (2^2)
constexpr int kPqSegs = (kPqExpHi - kPqExpLo) * 128;

__device__ __forceinline__ float fp_tf_pq_tab(float f, const float4* __restrict__ tab) {
    const uint32_t b = __builtin_bit_cast(uint32_t, f);
    const uint32_t idx = (b >> 16) - ((uint32_t)kPqExpLo << 7);  // sign bit set or exponent below the table: wraps to a huge value
    if (idx < (uint32_t)kPqSegs) {
        const float4 sg = tab[idx];
        const float xm = __builtin_bit_cast(float, (b & 0xFFFF0000u) | 0x00008000u);  // midpoint of the segment
        const float t = f - xm;
        return sg.x + __builtin_fmaf(t, __builtin_fmaf(t, sg.w, sg.z), sg.y);
    }
    // Outside the table. The two cases that are COMMON in images get their value directly, so that a wave with an out-of-gamut
    // (negative) or a black sample does not run the ~110 f64 operations of the general form (r3: the synthetic 8K PQ frames
    // have a negative linear sample in nearly every wave -- the "table" kernel executed the f64 form for all of them):
    //   f < 0, -inf, +inf, NaN: Math.pow(negative, 0.159...) is NaN (inf: inf / inf) and stays NaN to the end;
    //   f = +-0: pow = 0, so the result is (float)pow(0.8359375, 78.84375) = 0x354436E8.
    if ((b & 0x7FFFFFFFu) == 0u) return __builtin_bit_cast(float, 0x354436E8u);
    if (b >= 0x7F800000u) return __builtin_nanf("");  // sign bit set (and not -0), +inf, NaN
    return fp_tf_pq(f);  // (0, 2^-40) and [4, inf)
}

// ---- PQ + 16-bit quantisation, exact (r3) -------------------------------------------------------------------------------------
// TF_PQ.fromLinear followed by ImageBuffer.castToIntWithMax(65535) is a non-decreasing function float -> {0..65535}: thr[k] is
// the smallest float that reaches level k (host: build_pq16_thresholds, bisection over bit patterns with the reference's own
// formula; thr[0] = -inf, thr[65536] = +inf). The table form above gives the level to within +-1 (its float is within 1 ulp,
// i.e. 0.004 levels, of the reference's), and ONE comparison against the two neighbouring thresholds settles it: the
// oracle's integer (glibc pow) for every input (all 2^32 checked: tools/pq_sweep.py --pq16), where the float route alone differs by one
// level for 1 input in ~10^4.
__device__ __forceinline__ int fp_pq16(float f, const float4* __restrict__ tab, const float* __restrict__ thr) {
    if (!(f >= thr[1])) return 0;      // below the first threshold, negative, zero, NaN
    if (f >= thr[65535]) return f == __builtin_inff() ? 0 : 65535;  // (+inf: pow gives inf / inf = NaN, and (int)NaN is 0)
    const float t = fp_tf_pq_tab(f, tab);  // f is inside (0, 1): the table's range
    const float v = t * 65535.0f + 0.5f;
    int q = (int)v;
    // t is within 1 ulp of the reference's float (<= 0.004 levels), the two float roundings of v add <= 0.004: unless v lies
    // within 0.01 of an integer, q IS the reference's level and the thresholds need not be read (2 % of the samples do)
    const float fr = v - (float)q;
    if (fr < 0.01f || fr > 0.99f) {
        q = q < 1 ? 1 : q > 65534 ? 65534 : q;
        const float lo = thr[q], hi = thr[q + 1];
        q += (f >= hi ? 1 : 0) - (f < lo ? 1 : 0);
    }
    return q;
}

// ---- sRGB + 8-bit quantisation as a table of thresholds (r3) ---------------------------------------------------------------
// The composite the reference applies to a sample on its way into an 8-bit PNG -- TF_SRGB.fromLinearF (TransferFunction.java:39-44:
// one double pow, a float multiply and add) and ImageBuffer.castToIntWithMax(255) ((int)(t * 255 + 0.5f), clamped,
// ImageBuffer.java:129-147) -- is a non-decreasing function float -> {0..255}. So it is fully described by the smallest float
// that reaches each level: the host finds those 255 thresholds by bisection over float bit patterns with the reference's own
// formula (build_srgb8_table, host.hip) and files them by segment (the floats of [2^-9, 1) with equal bits >> 16: 9 binades x
// 128; a segment holds at most 3 thresholds). The device looks up base level + up to three comparisons instead of ~110 f64
// operations: the same integer for EVERY input by construction (checked over all 2^32 inputs: tools/pq_sweep.py --srgb8).
// Below 2^-9 the reference is on its linear branch (f * 12.92f): the same float operations are done here directly.
constexpr int kSrgb8ExpLo = 118;  // biased exponent of 2^-9
constexpr int kSrgb8Segs = (127 - kSrgb8ExpLo) * 128;

__device__ __forceinline__ int fp_srgb8(float f, const float4* __restrict__ tab) {
    const uint32_t b = __builtin_bit_cast(uint32_t, f);
    const uint32_t idx = (b >> 16) - ((uint32_t)kSrgb8ExpLo << 7);  // negative, tiny, >= 1, NaN: outside
    if (idx < (uint32_t)kSrgb8Segs) {
        const float4 sg = tab[idx];
        return (int)sg.x + (f >= sg.y ? 1 : 0) + (f >= sg.z ? 1 : 0) + (f >= sg.w ? 1 : 0);
    }
    if (f >= 1.0f) return 255;  // 1.055f * pow - 0.055f >= 1 - 2^-24: 255.49998 and up, clamped
    if (!(f == f)) return 0;    // (int)NaN
    const float v = (f * 12.92f) * 255.0f + 0.5f;  // linear branch (f < 0.0031306684), Java (int) cast, clamp
    if (v >= 255.0f) return 255;
    if (!(v >= 0.0f)) return 0;  // negative (and -inf)
    return (int)v;
}

// ---- PQ + 8-bit quantisation, exact (r3): 255 thresholds, binary search (a rare output format: no segment table) -----------------
// thr[0] = -inf, thr[k] = the smallest float of level k, thr[256] = +inf (build_pq8_thresholds); +inf and NaN give 0 as in the
// reference (pow(inf) / pow(inf) = NaN, (int)NaN = 0).
__device__ __forceinline__ int fp_pq8(float f, const float* __restrict__ thr) {
    if (!(f >= thr[1]) || f == __builtin_inff()) return 0;
    int q = 0;
#pragma unroll
    for (int step = 128; step >= 1; step >>= 1)
        if (f >= thr[q + step]) q += step;  // q + step <= 255
    return q;
}

// ---- sRGB + 16-bit quantisation, exact (r3): the two ideas above combined -------------------------------------------------
// tab16: quadratic segments of T(x) = c1 x^(1/2.4) - c0 (c1, c0 the reference's float constants) over [2^-9, 1), laid out like
// the PQ table; thr: the 65 535 thresholds of TF_SRGB.fromLinearF + castToIntWithMax(65535) (build_srgb16_thresholds). The
// reference rounds three times on the way to its float (pow cast, multiply, add), the table once: the level guess can be off by
// one where t * 65535 + 0.5 lies within 0.02 of an integer (4 % of the samples), and only those read their two thresholds.
// Below the branch point the reference's own float operations are done directly.
__device__ __forceinline__ int fp_srgb16(float f, const float4* __restrict__ tab16, const float* __restrict__ thr) {
    if (!(f >= thr[1])) return 0;       // below the first threshold, negative, zero, NaN
    if (f >= thr[65535]) return 65535;  // (+inf too: pow gives inf, (int)inf clamps)
    if (f < 0.00313066844250063f) return (int)((f * 12.92f) * 65535.0f + 0.5f);  // linear branch: positive, below 2656
    const uint32_t b = __builtin_bit_cast(uint32_t, f);
    const float4 sg = tab16[(b >> 16) - ((uint32_t)kSrgb8ExpLo << 7)];  // f in [0.00313, thr[65535]) inside [2^-9, 1)
    const float xm = __builtin_bit_cast(float, (b & 0xFFFF0000u) | 0x00008000u);
    const float tt = f - xm;
    const float t = sg.x + __builtin_fmaf(tt, __builtin_fmaf(tt, sg.w, sg.z), sg.y);
    const float v = t * 65535.0f + 0.5f;
    int q = (int)v;
    const float fr = v - (float)q;
    if (fr < 0.02f || fr > 0.98f) {
        q = q < 1 ? 1 : q > 65534 ? 65534 : q;
        const float lo = thr[q], hi = thr[q + 1];
        q += (f >= hi ? 1 : 0) - (f < lo ? 1 : 0);
    }
    return q;
}

// TF_SRGB.fromLinearF (TransferFunction.java:39-44)
__device__ __forceinline__ float fp_tf_srgb(float f) {
    if (f < 0.00313066844250063f) return f * 12.92f;
    return 1.055f * (float)fp_pow((double)f, 0.4166666666666667) + -0.055f;
}

}  // namespace jxl
