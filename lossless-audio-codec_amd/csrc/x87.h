// x87.h -- bit-exact software model of x87 80-bit extended arithmetic (64-bit significand, round to
// nearest even) for the on-device Levinson-Durbin solve.
//
// Why: the reference solves the LPC normal equations in `long double` (src/codec/lpc/lpc.cpp:98-154),
// which on its x86-64 build is the x87 extended format; the Q15 coefficients it emits -- and so the
// .lac bytes -- depend on that arithmetic.  CDNA4 has no such type, so the kernel carries its own.
// Scope: finite, normal values only (what the recursion can produce for |R| < 2^62); no NaN/Inf/
// denormal handling.  Every operation rounds once, exactly like fadd/fsub/fmul/fdiv under the default
// x87 control word (precision control = extended, RC = nearest).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define LACX_HD __host__ __device__ inline
#else
#define LACX_HD inline
#endif

namespace lacx {

struct xf80 {
    uint64_t m;  // significand, bit 63 set unless the value is zero
    int32_t e;   // value = (-1)^s * m * 2^(e-63)
    uint32_t s;  // sign
};

LACX_HD int clz64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)v);
#else
    return v ? __builtin_clzll(v) : 64;
#endif
}

LACX_HD void mul64x64(uint64_t a, uint64_t b, uint64_t& hi, uint64_t& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    lo = a * b;
    hi = __umul64hi(a, b);
#else
    const unsigned __int128 p = (unsigned __int128)a * b;
    lo = (uint64_t)p;
    hi = (uint64_t)(p >> 64);
#endif
}

LACX_HD xf80 xf_zero() { return xf80{0, -(1 << 20), 0}; }

LACX_HD xf80 xf_from_i64(int64_t v) {
    if (v == 0) return xf_zero();
    xf80 r;
    r.s = v < 0;
    uint64_t a = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
    const int lz = clz64(a);
    r.m = a << lz;
    r.e = 63 - lz;
    return r;
}

LACX_HD xf80 xf_neg(xf80 a) {
    a.s ^= 1u;
    return a;
}

// All four operations below are written without data-dependent branches: one lane of the device kernel runs one
// recursion, and 64 recursions that disagree about "same sign or not", "how far apart are the exponents", "carry or
// not" would otherwise walk through every side of every branch one after the other.

// Round (hi, guard/sticky from lo + extra sticky) to nearest even.
LACX_HD xf80 xf_round(uint32_t s, int32_t e, uint64_t hi, uint64_t lo, bool sticky) {
    const uint64_t guard = lo >> 63;
    const uint64_t st = (sticky || (lo << 1) != 0) ? 1u : 0u;
    const uint64_t inc = guard & (st | (hi & 1u));
    hi += inc;
    const bool ovf = inc != 0 && hi == 0;  // carried out of bit 63
    hi = ovf ? 0x8000000000000000ull : hi;
    e += ovf ? 1 : 0;
    return xf80{hi, e, s};
}

LACX_HD xf80 xf_mul(xf80 a, xf80 b) {
    uint64_t hi, lo;
    mul64x64(a.m, b.m, hi, lo);
    const bool low = (hi >> 63) == 0;  // product in [2^126, 2^127): normalise by one bit
    hi = low ? (hi << 1) | (lo >> 63) : hi;
    lo = low ? lo << 1 : lo;
    xf80 r = xf_round(a.s ^ b.s, a.e + b.e + (low ? 0 : 1), hi, lo, false);
    const bool zero = a.m == 0 || b.m == 0;
    r.m = zero ? 0 : r.m;
    r.e = zero ? -(1 << 20) : r.e;
    return r;
}

LACX_HD xf80 xf_add(xf80 a, xf80 b) {
    // x = the operand of larger magnitude (a zero has the smallest exponent there is)
    const bool swap = b.e > a.e || (b.e == a.e && b.m > a.m);
    xf80 x, y;
    x.m = swap ? b.m : a.m;
    x.e = swap ? b.e : a.e;
    x.s = swap ? b.s : a.s;
    y.m = swap ? a.m : b.m;
    y.e = swap ? a.e : b.e;
    y.s = swap ? a.s : b.s;
    // (yh:yl) = (y.m:0) >> d, sticky = whether ones were shifted out below bit 0 of yl
    uint32_t d = (uint32_t)(x.e - y.e);
    d = d > 128u ? 128u : d;
    const uint32_t dl = d & 63u;
    const uint64_t sh_r = y.m >> dl;
    const uint64_t sh_l = (y.m << 1) << (63u - dl);  // y.m << (64 - dl); 0 for dl == 0
    const bool lo_half = d < 64u, top = d >= 128u;
    const uint64_t yh = lo_half ? sh_r : 0;
    const uint64_t yl = lo_half ? sh_l : (top ? 0 : sh_r);
    const bool sticky = !lo_half && (top ? y.m != 0 : sh_l != 0);
    // same signs: (x.m:0) + (yh:yl), one bit to the right on a carry out of bit 127
    uint64_t ah = x.m + yh, al = yl;
    const bool carry = ah < x.m;
    const bool st_add = sticky || (carry && (al & 1u));
    al = carry ? (al >> 1) | (ah << 63) : al;
    ah = carry ? (ah >> 1) | 0x8000000000000000ull : ah;
    // different signs: (x.m:0) - (yh:yl) - (sticky ? 1 : 0), then to the left until bit 127 is set
    const uint64_t stb = sticky ? 1u : 0u;
    uint64_t sl = (uint64_t)0 - yl;
    uint64_t sh = x.m - yh - (yl != 0 ? 1u : 0u) - (sl < stb ? 1u : 0u);
    sl -= stb;
    const bool cancelled = (sh | sl) == 0;
    const uint32_t lz = sh ? (uint32_t)clz64(sh) : 64u + (uint32_t)clz64(sl);  // 128 when cancelled (not used then)
    const uint32_t l = lz & 63u;
    const uint64_t nh = lz < 64u ? (sh << l) | ((sl >> 1) >> (63u - l)) : sl << l;
    const uint64_t nl = lz < 64u ? sl << l : 0;
    const bool sub = x.s != y.s;
    xf80 r = xf_round(x.s, sub ? x.e - (int32_t)lz : x.e + (carry ? 1 : 0), sub ? nh : ah, sub ? nl : al,
                      sub ? sticky : st_add);
    const bool zero_out = sub && cancelled;  // exact cancellation -> +0
    r.m = zero_out ? 0 : r.m;
    r.e = zero_out ? -(1 << 20) : r.e;
    r.s = zero_out ? 0u : r.s;
    const bool pass = y.m == 0;  // nothing to add: the other operand as it is (a when both are zero)
    r.m = pass ? x.m : r.m;
    r.e = pass ? x.e : r.e;
    r.s = pass ? x.s : r.s;
    return r;
}

LACX_HD xf80 xf_sub(xf80 a, xf80 b) { return xf_add(a, xf_neg(b)); }

// floor((uh * 2^64 + ul) / D) and the remainder, for a 96-bit dividend below D * 2^32 (so the quotient fits 32 bits) and
// a divisor with bit 63 set.  The quotient is estimated in double precision -- every rounding on the way is below
// 2^-52 relative, so the estimate of a 32-bit quotient is off by far less than one -- and corrected by one unit either
// way from the exact 128-bit remainder.
LACX_HD uint32_t div96by64(uint32_t uh, uint64_t ul, uint64_t D, double rcp_d, uint64_t* rem) {
    const double num = (double)uh * 18446744073709551616.0 + (double)ul;
    double est = num * rcp_d;  // rcp_d = 1 / (double)D, rounded
    est = est < 4294967295.0 ? est : 4294967295.0;
    uint64_t q = (uint64_t)est;
    // r = dividend - q * D  (q * D has 96 bits)
    const uint64_t dlo = D & 0xFFFFFFFFull, dhi = D >> 32;
    const uint64_t p0 = q * dlo, p1 = q * dhi;             // each < 2^64
    const uint64_t pl = p0 + (p1 << 32);
    const uint64_t ph = (p1 >> 32) + (pl < p0 ? 1u : 0u);
    uint64_t rl = ul - pl;
    int64_t rh = (int64_t)((uint64_t)uh - ph - (ul < pl ? 1u : 0u));
    const bool neg = rh < 0;  // estimate one too high
    {
        const uint64_t t = rl + D;
        rh += (neg && t < rl) ? 1 : 0;
        rl = neg ? t : rl;
        q -= neg ? 1u : 0u;
    }
    const bool ge = rh != 0 || rl >= D;  // estimate one too low
    rl -= ge ? D : 0;
    q += ge ? 1u : 0u;
    *rem = rl;
    return (uint32_t)q;
}

LACX_HD xf80 xf_div(xf80 a, xf80 b) {  // b != 0
    // quotient significand Q = floor(a.m * 2^(63 + t) / b.m) in [2^63, 2^64), t = [a.m < b.m]
    const bool t = a.m < b.m;
    const uint64_t nh = t ? a.m : a.m >> 1, nl = t ? 0 : a.m << 63;  // the 128-bit dividend
    uint64_t r1, rem;
    const double rcp_d = 1.0 / (double)b.m;
    const uint32_t q1 = div96by64((uint32_t)(nh >> 32), (nh << 32) | (nl >> 32), b.m, rcp_d, &r1);
    const uint32_t q0 = div96by64((uint32_t)(r1 >> 32), (r1 << 32) | (nl & 0xFFFFFFFFull), b.m, rcp_d, &rem);
    uint64_t q = ((uint64_t)q1 << 32) | q0;
    int32_t e = a.e - b.e - (t ? 1 : 0);
    // round to nearest even on the remainder: compare 2 * rem with b.m
    const uint64_t other = b.m - rem;  // > 0
    const bool up = rem != 0 && (rem > other || (rem == other && (q & 1u)));
    q += up ? 1u : 0u;
    const bool ovf = up && q == 0;
    q = ovf ? 0x8000000000000000ull : q;
    e += ovf ? 1 : 0;
    const bool zero = a.m == 0;
    return xf80{zero ? 0 : q, zero ? -(1 << 20) : e, a.s ^ b.s};
}

// a < b
LACX_HD bool xf_lt(xf80 a, xf80 b) {
    // a zero has the smallest exponent and significand there are, and counts as positive whatever its sign bit says
    const bool mag_lt = a.e < b.e || (a.e == b.e && a.m < b.m);  // |a| < |b|
    const bool mag_gt = b.e < a.e || (a.e == b.e && b.m < a.m);
    const bool na = a.s != 0 && a.m != 0, nb = b.s != 0 && b.m != 0;
    return na != nb ? na : (na ? mag_gt : mag_lt);
}

// static_cast<double>(a) then std::round(c * 32768.0), clamped to int16
// (src/codec/lpc/lpc.cpp:73-78,179): two roundings, reproduced in integer arithmetic.
LACX_HD int16_t xf_to_q15(xf80 a) {
    if (a.m == 0) return 0;
    // 1) round the 64-bit significand to 53 bits (nearest even): double conversion
    uint64_t m53 = a.m >> 11;
    const uint64_t low = a.m & 0x7FFu;
    int32_t e = a.e;
    if (low > 0x400u || (low == 0x400u && (m53 & 1u))) {
        ++m53;
        if (m53 >> 53) {
            m53 >>= 1;
            ++e;
        }
    }
    // value = m53 * 2^(e-52); times 2^15 -> m53 * 2^(e-37)
    const int32_t sh = e - 37;
    int64_t mag;
    if (sh >= 11) {  // >= 2^63 after scaling: far outside int16
        mag = 40000;
    } else if (sh >= 0) {
        const uint64_t v = m53 << sh;
        mag = v > 40000u ? 40000 : (int64_t)v;
    } else if (sh < -54) {
        mag = 0;  // < 0.5
    } else {
        const int r = -sh;  // 1..54
        const uint64_t ip = (r >= 64) ? 0 : (m53 >> r);
        const uint64_t half = (m53 >> (r - 1)) & 1u;  // first fractional bit
        const uint64_t v = ip + half;                // round half away from zero (on the magnitude)
        mag = v > 40000u ? 40000 : (int64_t)v;
    }
    int64_t q = a.s ? -mag : mag;
    if (q < -32768) q = -32768;
    if (q > 32767) q = 32767;
    return (int16_t)q;
}

// 0.999L and 1e-8L as the compiler rounds them to extended precision (values checked in
// tests/test_x87.py against the host's long double).
LACX_HD xf80 xf_const_0_999() { return xf80{0xFFBE76C8B4395810ull, -1, 0}; }
LACX_HD xf80 xf_const_1em8() { return xf80{0xABCC77118461CEFDull, -27, 0}; }
LACX_HD xf80 xf_const_one() { return xf80{0x8000000000000000ull, 0, 0}; }

// Levinson-Durbin exactly as src/codec/lpc/lpc.cpp:98-154 + analyze_block_q15 :156-186, for all the
// encoder's candidate orders at once: the loop body for iteration i does not depend on the target
// order, so the order-c solve is the order-12 solve stopped after c iterations.
// r[0..12]: exact integer autocorrelation. max_order: highest candidate to produce (<= 12).
// out_coef[ci][0..12], out_used[ci] for ci over cands {4,6,8,10,12}.
//
// The three 13-element work arrays are reached through an accessor so that the device kernel can keep them in
// LDS (one column per thread): indexed by loop variables, plain local arrays would live in scratch memory and
// every access would pay a trip through the vector memory path.
struct XfLocalArray {
    xf80 v[13];
    LACX_HD xf80 get(int i) const { return v[i]; }
    LACX_HD void set(int i, xf80 x) { v[i] = x; }
};

template <class Arr, class GetR, class PutCoef, class PutUsed>
LACX_HD void levinson_candidates_t(GetR&& get_r, int max_valid_order, Arr& R, Arr& a, Arr& prevA, PutCoef&& put_coef,
                                   PutUsed&& put_used) {
    const xf80 eps = xf_const_1em8();
    const xf80 lim = xf_const_0_999();
    const xf80 one = xf_const_one();
    for (int i = 0; i <= 12; ++i) {
        R.set(i, xf_from_i64(get_r(i)));
        a.set(i, xf_zero());
        prevA.set(i, xf_zero());
    }
    if (xf_lt(R.get(0), one)) R.set(0, one);  // lpc.cpp:169-172 (energy < 1 -> 1)
    xf80 E = R.get(0);
    int achieved = 0;
    bool stopped = xf_lt(E, eps);  // lpc.cpp:105-109 (cannot trigger after the clamp to 1)
    int next_ci = 0;
    for (int i = 1; i <= 12; ++i) {
        if (!stopped) {
            xf80 acc = xf_zero();
            for (int j = 1; j < i; ++j) acc = xf_add(acc, xf_mul(prevA.get(j), R.get(i - j)));
            if (xf_lt(E, eps)) {
                stopped = true;
            } else {
                xf80 ki = xf_div(xf_sub(R.get(i), acc), E);
                if (xf_lt(lim, ki)) ki = lim;
                if (xf_lt(ki, xf_neg(lim))) ki = xf_neg(lim);
                const xf80 e_new = xf_mul(xf_sub(one, xf_mul(ki, ki)), E);
                if (xf_lt(e_new, eps)) {
                    stopped = true;
                } else {
                    a.set(i, ki);
                    for (int j = 1; j < i; ++j) a.set(j, xf_sub(prevA.get(j), xf_mul(ki, prevA.get(i - j))));
                    for (int j = 1; j <= i; ++j) prevA.set(j, a.get(j));
                    E = e_new;
                    achieved = i;
                }
            }
        }
        // candidate order == i: snapshot (used = min(i, achieved))
        if (next_ci < 5 && i == 4 + 2 * next_ci) {
            const bool keep = i <= max_valid_order;
            put_used(next_ci, keep ? (uint8_t)achieved : (uint8_t)0);
            for (int j = 0; j <= 12; ++j)
                put_coef(next_ci, j, (keep && j >= 1 && j <= achieved) ? xf_to_q15(a.get(j)) : (int16_t)0);
            ++next_ci;
        }
    }
}

LACX_HD void levinson_candidates(const int64_t* r, int max_valid_order, int16_t out_coef[5][13],
                                 uint8_t out_used[5]) {
    XfLocalArray R, a, prevA;
    levinson_candidates_t(
        [r](int i) { return r[i]; }, max_valid_order, R, a, prevA,
        [out_coef](int ci, int j, int16_t v) { out_coef[ci][j] = v; }, [out_used](int ci, uint8_t v) { out_used[ci] = v; });
}

}  // namespace lacx
