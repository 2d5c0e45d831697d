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

// Round (hi, guard/sticky from lo + extra sticky) to nearest even.
LACX_HD xf80 xf_round(uint32_t s, int32_t e, uint64_t hi, uint64_t lo, bool sticky) {
    const bool guard = (lo >> 63) != 0;
    const bool st = sticky || ((lo << 1) != 0);
    if (guard && (st || (hi & 1u))) {
        ++hi;
        if (hi == 0) {
            hi = 0x8000000000000000ull;
            ++e;
        }
    }
    return xf80{hi, e, s};
}

LACX_HD xf80 xf_mul(xf80 a, xf80 b) {
    if (a.m == 0 || b.m == 0) {
        xf80 z = xf_zero();
        z.s = a.s ^ b.s;
        return z;
    }
    uint64_t hi, lo;
    mul64x64(a.m, b.m, hi, lo);
    int32_t e = a.e + b.e + 1;
    if (!(hi >> 63)) {  // product in [2^126, 2^127): normalise by one bit
        hi = (hi << 1) | (lo >> 63);
        lo <<= 1;
        e -= 1;
    }
    return xf_round(a.s ^ b.s, e, hi, lo, false);
}

LACX_HD xf80 xf_add(xf80 a, xf80 b) {
    if (b.m == 0) return a;
    if (a.m == 0) return b;
    // make |a| >= |b|
    if (b.e > a.e || (b.e == a.e && b.m > a.m)) {
        const xf80 t = a;
        a = b;
        b = t;
    }
    const int64_t d = (int64_t)a.e - (int64_t)b.e;
    uint64_t bh, bl;
    bool sticky = false;
    if (d == 0) {
        bh = b.m;
        bl = 0;
    } else if (d < 64) {
        bh = b.m >> d;
        bl = b.m << (64 - d);
    } else if (d == 64) {
        bh = 0;
        bl = b.m;
    } else if (d < 128) {
        bh = 0;
        bl = b.m >> (d - 64);
        sticky = (b.m << (128 - d)) != 0;
    } else {
        bh = 0;
        bl = 0;
        sticky = true;
    }
    uint64_t rh, rl;
    int32_t e = a.e;
    if (a.s == b.s) {
        rl = bl;  // a's low half is zero
        rh = a.m + bh;
        if (rh < a.m) {  // carry out of bit 127
            sticky = sticky || (rl & 1u);
            rl = (rl >> 1) | (rh << 63);
            rh = (rh >> 1) | 0x8000000000000000ull;
            ++e;
        }
        return xf_round(a.s, e, rh, rl, sticky);
    }
    // magnitude subtraction: (a.m:0) - (bh:bl) - (sticky ? 1 : 0)
    rl = (uint64_t)0 - bl;
    uint64_t borrow = bl != 0;
    rh = a.m - bh - borrow;
    if (sticky) {
        if (rl == 0) --rh;
        --rl;
    }
    if (rh == 0 && rl == 0) return xf_zero();  // exact cancellation -> +0
    int lz;
    if (rh) {
        lz = clz64(rh);
        if (lz) {
            rh = (rh << lz) | (rl >> (64 - lz));
            rl <<= lz;
        }
    } else {
        lz = 64 + clz64(rl);
        rh = rl << (lz - 64);
        rl = 0;
    }
    e -= lz;
    return xf_round(a.s, e, rh, rl, sticky);
}

LACX_HD xf80 xf_sub(xf80 a, xf80 b) { return xf_add(a, xf_neg(b)); }

LACX_HD xf80 xf_div(xf80 a, xf80 b) {  // b != 0
    if (a.m == 0) {
        xf80 z = xf_zero();
        z.s = a.s ^ b.s;
        return z;
    }
    uint64_t q, rem;
    int32_t e = a.e - b.e;
    int iters;
    if (a.m >= b.m) {
        q = 1;
        rem = a.m - b.m;
        iters = 63;
    } else {
        q = 0;
        rem = a.m;
        iters = 64;
        e -= 1;
    }
    for (int i = 0; i < iters; ++i) {
        const bool carry = (rem >> 63) != 0;
        rem <<= 1;
        if (carry || rem >= b.m) {
            rem -= b.m;
            q = (q << 1) | 1u;
        } else {
            q <<= 1;
        }
    }
    // round to nearest even on the remainder: compare 2*rem with b.m
    bool up = false;
    if (rem != 0) {
        const uint64_t other = b.m - rem;  // > 0
        if (rem > other) {
            up = true;
        } else if (rem == other) {
            up = (q & 1u) != 0;
        }
    }
    if (up) {
        ++q;
        if (q == 0) {
            q = 0x8000000000000000ull;
            ++e;
        }
    }
    return xf80{q, e, a.s ^ b.s};
}

// a < b
LACX_HD bool xf_lt(xf80 a, xf80 b) {
    const bool az = a.m == 0, bz = b.m == 0;
    if (az && bz) return false;
    if (az) return b.s == 0;
    if (bz) return a.s != 0;
    if (a.s != b.s) return a.s != 0;
    bool mag_lt;  // |a| < |b|
    bool mag_eq = false;
    if (a.e != b.e) {
        mag_lt = a.e < b.e;
    } else if (a.m != b.m) {
        mag_lt = a.m < b.m;
    } else {
        mag_lt = false;
        mag_eq = true;
    }
    if (mag_eq) return false;
    return a.s ? !mag_lt : mag_lt;
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
