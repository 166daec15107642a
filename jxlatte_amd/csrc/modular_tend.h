// ModularChannel.tendency (J/frame/modular/ModularChannel.java:23-47) and one pair of the inverse squeeze recurrence
// (:361-413) for the gfx950 kernels (k_modular.hip, k_modular_vh.hip). int32 arithmetic wraps as in Java: adds / muls in
// uint32, `/` truncates toward zero.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace jxl {

__device__ __forceinline__ int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__device__ __forceinline__ int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
__device__ __forceinline__ int32_t wshl(int32_t a, int s) { return (int32_t)((uint32_t)a << s); }

// ---- the reference's form, branch-free, exact for every int32 triple ---------------------------------------------------------
// Split so that only the part that depends on `a` (the previously OUTPUT sample: the serial dependency of the squeeze
// recurrence) sits on the critical path; everything derived from b and c alone is prepared ahead (TendPre).
struct TendPre {
    int32_t base;  // -3c - b
    int32_t e;     // 2 (b - c)
    int32_t twob;  // 2 b
    int32_t b;
    bool ge, le;   // b >= c, b <= c
};

__device__ __forceinline__ TendPre tend_pre(int32_t b, int32_t c) {
    TendPre t;
    t.base = wsub(wmul(-3, c), b);
    t.e = wmul(2, wsub(b, c));
    t.twob = wmul(2, b);
    t.b = b;
    t.ge = b >= c;
    t.le = b <= c;
    return t;
}

__device__ __forceinline__ int32_t tend_apply(int32_t a, const TendPre& t) {
    const bool dec = t.ge && a >= t.b;           // if (a >= b && b >= c)
    const bool inc = !dec && t.le && a <= t.b;   // else if (a <= b && b <= c)
    const int32_t x = wadd(wadd(wmul(4, a), t.base), dec ? 6 : -6) / 12;
    const int32_t d = wsub(wmul(2, a), t.twob);
    // decreasing: if (x - (x&1) > d) x = d + 1; if (x + (x&1) > e) x = e;
    int32_t xd = x;
    xd = wsub(xd, xd & 1) > d ? wadd(d, 1) : xd;
    xd = wadd(xd, xd & 1) > t.e ? t.e : xd;
    // increasing: if (x + (x&1) < d) x = d - 1; if (x - (x&1) < e) x = e;
    int32_t xi = x;
    xi = wadd(xi, xi & 1) < d ? wsub(d, 1) : xi;
    xi = wsub(xi, xi & 1) < t.e ? t.e : xi;
    return dec ? xd : (inc ? xi : 0);
}

__device__ __forceinline__ int32_t tendency(int32_t a, int32_t b, int32_t c) { return tend_apply(a, tend_pre(b, c)); }

// ---- the sign-normalised short form (r5) -------------------------------------------------------------------------------------
// tendency is odd -- tendency(-a, -b, -c) = -tendency(a, b, c): the increasing branch is the mirror image of the decreasing
// one, truncating division included -- and it is 0 whenever b == c. So every triple is folded onto the decreasing branch
// with m = (b < c) ? -1 : 0, v' = (v ^ m) - m:
//     valid = a' >= b' (>= c' by construction)
//     x     = (4a' - 3c' - b' + 6) / 12      numerator >= 6 when valid: an unsigned multiply-high, no sign fix-up
//     "if (x - (x&1) > d) x = d + 1; if (x + (x&1) > e) x = e"  ==  min(x, d + 1, e)      d = 2(a'-b'), e = 2(b'-c') both even
//     (r2 identity: comparing the even number below / above x with an even bound is comparing x with the bound +-1)
//     not valid: d + 1 <= -1 and x >= 0 (a logical shift of an unsigned product), e >= 0, so max(min3(..), 0) = 0 selects
//     nothing: no comparison, no select
//     result = (r ^ m) - m, whose "- m" is folded into the residual ahead of the chain
// 13 dependent instructions per pair and ~14 ahead of the chain, against ~37 + ~18 of the two-candidate form of r2-r4
// (SQ_INSTS_VALU 64 per pair and lane, profiles/r5_modular_*). The identities need no intermediate to wrap: guaranteed while
// |avg|, |next avg|, |residual| < 2^23 and |left| < 2^27 -- then every output is below 1.07 * 2^26 again (|tendency| <=
// 2|b - c| <= 2^25), so a chunk whose inputs pass squeeze_range_ok stays inside by induction. Chunks that do not pass are
// walked with the exact form above. tests/test_oracle_kats.py::test_tendency_normalised_form restates this instruction by
// instruction in numpy (exhaustive on [-40, 40]^3, random and extreme triples of the guarded range) against the reference.
struct TendN {
    int32_t m;     // -1 if b < c else 0
    int32_t base;  // 3|b - c| + 6 - 4b'  (= -3c' - b' + 6)
    int32_t k1;    // 1 - 2b'
    int32_t e2;    // 2|b - c|
    int32_t rm;    // residual - m
    int32_t b;     // avg (un-normalised): first = b + diff / 2
};

__device__ __forceinline__ TendN tend_n_pre(int32_t b, int32_t c, int32_t res) {
    TendN t;
    const int32_t bmc = wsub(b, c);
    t.m = bmc >> 31;
    const int32_t ab = wsub(bmc ^ t.m, t.m);
    const int32_t bn = wsub(b ^ t.m, t.m);
    t.e2 = wshl(ab, 1);
    t.base = wsub(wadd(wadd(wshl(ab, 1), ab), 6), wshl(bn, 2));
    t.k1 = wsub(1, wshl(bn, 1));
    t.rm = wsub(res, t.m);
    t.b = b;
    return t;
}

// one pair: left -> (first, second); returns second (the next pair's left)
__device__ __forceinline__ int32_t squeeze_pair_n(int32_t left, const TendN& t, int32_t& first, int32_t& second) {
    const int32_t an = wsub(left ^ t.m, t.m);
    const uint32_t n = (uint32_t)wadd(wshl(an, 2), t.base);
    const int32_t x = (int32_t)(__umulhi(n, 0xAAAAAAABu) >> 3);  // n / 12, exact for every uint32
    const int32_t d1 = wadd(wshl(an, 1), t.k1);
    const int32_t r = max(min(min(x, d1), t.e2), 0);
    const int32_t diff = wadd(t.rm, r ^ t.m);
    first = wadd(t.b, diff / 2);
    second = wsub(first, diff);
    return second;
}

// the exact pair (any int32)
__device__ __forceinline__ int32_t squeeze_pair_exact(int32_t left, int32_t b, int32_t c, int32_t res, int32_t& first, int32_t& second) {
    const int32_t diff = wadd(res, tend_apply(left, tend_pre(b, c)));
    first = wadd(b, diff / 2);
    second = wsub(first, diff);
    return second;
}

// Range guard of the short form over a chunk: running max / min of the averages and residuals (one v_max3 + one v_min3 per
// pair) and the incoming left.
constexpr int32_t kSqueezeSafeIn = 1 << 23, kSqueezeSafeLeft = 1 << 27;
struct SqueezeRange {
    int32_t hi, lo;
    __device__ __forceinline__ void init(int32_t v) { hi = lo = v; }
    __device__ __forceinline__ void add(int32_t a, int32_t r) {
        hi = max(max(hi, a), r);
        lo = min(min(lo, a), r);
    }
    __device__ __forceinline__ bool ok(int32_t left) const {
        return hi < kSqueezeSafeIn && lo > -kSqueezeSafeIn && (uint32_t)(left + kSqueezeSafeLeft) < 2u * (uint32_t)kSqueezeSafeLeft;
    }
};

}  // namespace jxl
