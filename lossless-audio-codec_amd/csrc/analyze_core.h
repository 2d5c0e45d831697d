// analyze_core.h -- the per-thread phases of the channel-block analysis kernel.
//
// One workgroup analyses one "slot" (a channel-segment of n <= CH*T samples) and produces exactly the
// decisions Block::Encoder::encode makes before emitting bits (ref src/codec/block/encoder.cpp:313-552).
// The reference walks every residual serially with stateful Rice models; here the same numbers come
// from feed-forward array operations:
//   * thread t owns the CH consecutive samples [t*CH, t*CH+CH); CH divides 32, so the reference's
//     256-sample drift window and 96-sample micro window start on chunk boundaries and their sums are
//     differences of block-scanned chunk prefixes;
//   * the adaptive Rice parameter after sample j depends only on prefix sums of u = zigzag(residual)
//     (Rice::adapt_k never feeds its result back into its state, ref src/codec/rice/rice.hpp:45-114),
//     and is obtained without a division (kmean());
//   * all 16 static-k and 13 initial-k costs come from bit-plane population counts:
//     sum_j (u_j >> k) = sum_{b>=k} 2^(b-k) * C_b, with C_b counted by bit-sliced adders per thread,
//     wave ballots per block, and group prefix tables per partition (ref block/encoder.cpp:121-188);
//   * zero runs are resolved with a block prefix-max of "last non-zero index" and a 3-sample lookahead.
//
// Data layout in LDS: sample j of the slot lives at word sw(j) = (j mod CH) * T + (j div CH), i.e.
// element i of every thread's chunk forms one contiguous row of T words.  A wave reading "its" element
// i touches 64 consecutive words (conflict-free), and so do the window taps j-256 / j-96 (same row,
// 256/CH resp. 96/CH words to the left).
//
// The functions here are plain per-thread code (no cross-lane operations) so that the same text is
// compiled into the HIP kernel (k_analyze.hip) and into the lock-step host simulator used by the CPU
// tests (tests/native/sim_analyze.cpp).  Cross-thread steps (block scans, ballots, LDS atomics) live in
// the drivers.
#pragma once
#include <type_traits>

#include "lacx_types.h"
#include "x87.h"

namespace lacx {

template <int CH_, int T_>
struct Geo {
    static constexpr int CH = CH_;
    static constexpr int T = T_;
    static constexpr int MAXN = CH_ * T_;
    static constexpr int LV = (CH_ == 16) ? 5 : ((CH_ == 8) ? 4 : 3);  // bits of a count in 0..CH
    static constexpr int TPG = 64 / CH_;      // threads per 64-sample group
    static constexpr int NG = MAXN / 64;      // 64-sample groups
    static constexpr int W256 = 256 / CH_;    // drift window, in chunks
    static constexpr int W96 = 96 / CH_;      // micro window, in chunks
    static constexpr int MAXP = (MAXN >= 8192) ? 8 : ((MAXN >= 4096) ? 7 : ((MAXN >= 2048) ? 6 : ((MAXN >= 1024) ? 5 : ((MAXN >= 512) ? 4 : 3))));
    static constexpr int NSEG = (2 << MAXP) - 2;  // segments over all partition orders 1..MAXP
    static_assert(32 % CH_ == 0, "chunk must divide the 96/256 windows");
    static_assert(MAXN % 64 == 0, "whole groups");
};

// transposed LDS index of sample j (j >= 0)
template <class G>
LACX_HD int sw(int j) {
    return (j % G::CH) * G::T + (j / G::CH);
}

// Where a slot's samples come from: a plain channel, or M/S derived on the fly
// (ref src/codec/simd/neon.cpp:14-30: M = (L+R)>>1 arithmetic, S = L-R).  Three source layouts:
//   PCM_PLANAR_I32       a = left, b = right, int32 per sample (the reference's API boundary)
//   PCM_INTERLEAVED_I16  a = WAV data chunk: little-endian int16, frames interleaved (L R L R ...)
//   PCM_INTERLEAVED_I24  a = WAV data chunk: packed 3-byte little-endian samples, interleaved
enum : int { PCM_PLANAR_I32 = 0, PCM_INTERLEAVED_I16 = 1, PCM_INTERLEAVED_I24 = 2 };

struct SlotSrc {
    const int32_t* a;
    const int32_t* b;
    int kind;  // CH_L / CH_R / CH_M / CH_S
    int layout = PCM_PLANAR_I32;
    int channels = 2;
};

LACX_HD int32_t sext24(uint32_t v) { return (int32_t)(v << 8) >> 8; }
LACX_HD int32_t slot_combine(int kind, int32_t l, int32_t r);

LACX_HD int32_t slot_fetch(const SlotSrc& s, int64_t idx) {
    int32_t l, r = 0;
    if (s.layout == PCM_PLANAR_I32) {
        if (s.kind == CH_L) return s.a[idx];
        if (s.kind == CH_R) return s.b[idx];
        l = s.a[idx];
        r = s.b[idx];
    } else if (s.layout == PCM_INTERLEAVED_I16) {
        if (s.channels == 2) {
            const uint32_t w = reinterpret_cast<const uint32_t*>(s.a)[idx];  // one 4-byte load per frame
            l = (int32_t)(int16_t)(w & 0xFFFFu);
            r = (int32_t)(int16_t)(w >> 16);
        } else {
            l = reinterpret_cast<const int16_t*>(s.a)[idx];
        }
    } else {
        const uint8_t* p = reinterpret_cast<const uint8_t*>(s.a) + idx * 3 * s.channels;
        l = sext24((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16));
        if (s.channels == 2) r = sext24((uint32_t)p[3] | ((uint32_t)p[4] << 8) | ((uint32_t)p[5] << 16));
    }
    return slot_combine(s.kind, l, r);
}

LACX_HD uint32_t zigzag32(int32_t r) { return ((uint32_t)r << 1) ^ (uint32_t)(r >> 31); }

// Optimisation barrier on a per-thread value (device builds): what is derived from the result cannot be hoisted
// out of an enclosing loop.  Used where hoisting only lengthens live ranges and, at 128 VGPRs, ends in spills.
LACX_HD int opaque_i32(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

LACX_HD int ctz32(uint32_t v) {  // v != 0
    return __builtin_ctz(v);
}

LACX_HD int clz32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clz((int)v);
#else
    return v ? __builtin_clz(v) : 32;
#endif
}

// Rice parameter of Rice::adapt_k / adapt_k_stateless before biasing:
//   mean = (S + (c>>1)) / c ; k = mean <= 1 ? 0 : bit_width(mean-1)       (ref rice.hpp:68-71,
//   block/encoder.cpp:72-77) == the smallest k with mean <= 2^k, i.e. with (X - c) < (c << k), X = S + (c>>1).
LACX_HD uint32_t kmean(uint64_t S, uint32_t c) {
    const uint64_t X = S + (c >> 1);
    const uint64_t Y = X - c;  // >= c when the mean exceeds 1 (garbage otherwise: selected away below, no branch)
    const int g = clz64((uint64_t)c) - clz64(Y);  // bit_width(Y) - bit_width(c) >= 0
    const uint32_t k = (uint32_t)g + ((Y >> (g & 63)) >= c ? 1u : 0u);
    return (X < 2ull * c) ? 0u : k;
}
// same, for S + (c>>1) < 2^32
// (k <= 30 whatever the sums are: the mean never exceeds the largest u, and u < 2^30)
LACX_HD uint32_t kmean32(uint32_t S, uint32_t c) {
    const uint32_t X = S + (c >> 1);
    const uint32_t Y = X - c;
    const int g = clz32(c) - clz32(Y);
    const uint32_t k = (uint32_t)g + ((Y >> (g & 31)) >= c ? 1u : 0u);
    return (X < 2u * c) ? 0u : k;
}

// A block (a candidate's residual) is "narrow" when its sum of u stays below this: every prefix sum, every sum plus half a
// count, and every cost total of a partition (at most 36 bits of overhead per sample on top of u: 16384 x 36 < 2^20)
// then fits 32 bits -- all but the loudest 24-bit material.
constexpr uint64_t kNarrowLimit = (1ull << 32) - (1ull << 20);

template <bool NARROW>
LACX_HD uint32_t kmean_t(uint64_t S, uint32_t c) {
    return NARROW ? kmean32((uint32_t)S, c) : kmean(S, c);
}

LACX_HD uint64_t rice_cost(uint32_t u, uint32_t k) {  // ref block/encoder.cpp:67-70
    return (uint64_t)((k >= 31u) ? 0u : (u >> k)) + 1u + k;
}

// Bias of Rice::adapt_k (ref rice.hpp:83-113) applied to the unbiased km after `c` samples:
//   P = sum of u over [0,c), W = sum over [0, c-256), flag counts d = large | zero<<16 over the last 96.
// Division-free: with mean = floor(X/c), X = P + (c>>1):
//   3L > 4 mean  <=>  X < ceil(3L/4) * c          4L+3 < 3 mean  <=>  X >= (floor((4L+3)/3) + 1) * c
// 0 / 1 in a vector register.  On the device the value is pinned there: flags that are combined as integers stay on the
// vector unit.  Left to itself the compiler keeps comparison results as lane masks in scalar registers, combines them
// with scalar instructions and turns selects between costly operands into divergent branches -- each a round trip
// vector compare -> scalar unit -> execution mask, which is what the adaptive-cost loops spent a third of their time on.
LACX_HD uint32_t flag01(bool b) {
    uint32_t v = b ? 1u : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// STEADY: the caller guarantees c > 256 (hence also c >= 96): every chunk but the first 256 / CH of a slot.
template <bool NARROW, bool STEADY = false>
LACX_HD uint32_t biased_k(uint32_t km, uint64_t P, uint64_t W, uint32_t d, uint32_t c) {
    // local mean of the last 256 (ref rice.hpp:85-87); u < 2^30 keeps L, U, D inside 32 bits
    const uint32_t L = NARROW ? ((((uint32_t)P - (uint32_t)W) + 128u) >> 8) : (uint32_t)(((P - W) + 128u) >> 8);
    const uint32_t U = (3u * L + 3u) >> 2;
    const uint32_t D = L + (L + 3u) / 3u + 1u;
    const uint64_t X = NARROW ? (uint64_t)((uint32_t)P + (c >> 1)) : P + (c >> 1);
    const uint64_t Uc = (uint64_t)U * c, Dc = (uint64_t)D * c;  // both products always: no branch around the second
    // drift (ref rice.hpp:88-95): U c <= D c, so "up" and "down" exclude each other
    uint32_t act = flag01(X >= (uint64_t)c);
    if (!STEADY) act &= flag01(c > 256u);
    const int32_t drift = (int32_t)(act & flag01(X < Uc)) - (int32_t)(act & flag01(X >= Dc));
    // micro window (ref rice.hpp:97-105): large*4 >= 96*3 <=> large >= 72 ; zero*5 >= 96*4 <=> zero >= 77
    const uint32_t large = d & 0xFFFFu, zero = d >> 16;
    uint32_t big = flag01(large >= 72u), sml = flag01(zero >= 77u);
    if (!STEADY) {
        const uint32_t m = flag01(c >= 96u);
        big &= m;
        sml &= m;
    }
    sml &= big ^ 1u;
    // big: min(drift + 1, 1); small: max(drift - 1, -1); neither: drift -- one clamp of drift + big - small
    int32_t bias = drift + (int32_t)big - (int32_t)sml;
    bias = bias < -1 ? -1 : (bias > 1 ? 1 : bias);
    int bk = (int)km + bias;
    bk = bk < 0 ? 0 : (bk > 31 ? 31 : bk);
    return (uint32_t)bk;
}

struct SegInfo {
    uint64_t sbits;  // static Rice bits at sk
    uint8_t ak;      // adaptive initial k (estimate_initial_k)
    uint8_t sk;      // static k (estimate_static_k)
    uint8_t pad[6];
};

// Partition-search scratch; aliases the staged samples, which are dead once the winning residual is in u[].
template <class G>
struct PartMem {
    uint32_t grp[15][G::NG + 1];  // packed (two 16-bit fields) plane counts per 64-sample group -> prefix
    SegInfo seginfo[G::NSEG];
    unsigned long long segacc[G::NSEG][3];
    uint32_t segrun[G::NSEG];
    uint8_t choice[G::NSEG];
    unsigned long long pbits[G::MAXP + 1];
    // partition_quick: (chunk, order) pairs whose Rice parameter is not provably constant over the chunk, left to
    // partition_slow_entry; entry = chunk | (order - 1) << 12.  One region of 64 * MAXP entries per wave.
    uint32_t qcount;  // (host simulator only)
    uint32_t wqcount[16];  // entries every wave has queued (the queued pairs are then walked by all waves in equal shares)
    uint16_t queue[G::T * G::MAXP];
};

// Output side of the device emit (emit_core.h): the bit tile and the Rice parameter in force per sample.  Aliases the
// staged samples / the partition scratch in the analysis kernel's image (whole-block geometry only).
constexpr int kEmitTileWords = 12288;  // 48 KiB output tile (393216 bits)
template <class G>
struct EmitOut {
    uint32_t obits[kEmitTileWords];  // output tile, big-endian bit order inside each word
    uint8_t kin[G::MAXN];            // Rice parameter in force per sample (transposed)
};
struct NoEmitOut {};

template <class G>
struct Smem {
    uint32_t u[G::MAXN + 4];  // zigzag residual of the current candidate (transposed), always plain (the micro-window flags live in tabZM / tabLM); +4: peek_u looks up to 3 samples past the slot
    union XP {
        int32_t x[G::MAXN];  // staged samples (transposed), zero beyond n
        PartMem<G> part;
        typename std::conditional<G::T == 1024, EmitOut<G>, NoEmitOut>::type o;  // fused emit (after the plan is final)
    } xp;
    uint64_t tabP[G::T + 1];  // in: chunk sums; after scan: exclusive prefix, [T] = total
    int32_t tabNZ[G::T + 1];  // in: last non-zero index in chunk (-1); after scan: exclusive prefix max
    uint32_t tabF[G::T + 1];  // packed micro-window flag counts of every chunk (phase A)
    uint16_t tabZM[G::T];     // phase A: bit i = sample i of the chunk has the "zero quotient" micro flag (q == 0)
    uint16_t tabLM[G::T];     // phase A: bit i = sample i of the chunk has the "large quotient" micro flag (q > 3)
    uint16_t tabUZ[G::T + 2]; // phase A: bit i = sample i of the chunk is zero ([T]: stays 0, "not a zero" past the slot)
    uint16_t bqueue[G::T];    // chunks whose adaptive costs need the walk (phase B), packed over the lanes of the waves
    uint32_t bqcount;
    uint32_t planeTot[2][32];    // per bit-plane population over the block (double buffered by candidate parity)
    uint32_t planeTot256[2][32]; // ... over the first min(256,n) samples
    unsigned long long acc[2][4];  // rice, bin, zr bits and has_run of the current candidate
    uint32_t lbacc[11][3];         // per candidate: block sums behind the pruning bound (pass1_bounds): sum of bit_width(u) + 1; zeros | fours << 16; run ends
    uint32_t has4[2];              // current candidate has a run of >= 4 zero residuals (zero-run mode possible)
    uint64_t wtotP[16];  // per-wave totals used by the block scans
    int32_t wtotZ[16];
    uint32_t wtotF[16];
    // running best candidate (written by thread 0)
    uint64_t best_bits, best_rice, best_zr, best_bin, best_static;
    uint32_t best_k0, best_sk, best_hasrun;
    int32_t best_cand;
    int32_t next_cand;  // candidate chosen for the next exact evaluation (-1: none left)
    uint64_t cand_key[11];  // per candidate: pruning bound * 16 + index (all ones: not available)
    uint32_t cur_k0;
    LpcSet lpc;
    // the finished plan (thread 0 writes it here, then it is copied out cooperatively) and, for the emit fused into the
    // whole-block analysis kernel, the fields the emit phases of emit_core.h read (same names as in EmitMem)
    ChannelPlan plan;
    int32_t tabNX[G::T + 1];  // first non-zero index per chunk -> exclusive suffix min
    int32_t wx[16];
    uint8_t part_mode_k[kMaxParts];
    uint32_t ptype, order, p, parts, cand, header_bits, payload_bytes;
    uint32_t err;
    uint32_t plan_any_zr;  // finalize_plan: some partition of the plan uses zero-run mode
    uint32_t tmpl_state, tmpl_pad[3];  // silent slots: the state word of the launch's SilentTemplate as thread 0 read it
};

template <class G>
struct Thread {
    int tid;
    int a;       // first sample of the chunk
    uint32_t n;  // samples in the slot
    int cnt;     // samples of the chunk inside the slot (0..CH)
    uint32_t cs[G::LV];  // bit-sliced per-plane counts of this chunk
    unsigned long long crice, cbin, czr;  // chunk partial costs
    uint32_t chasrun;
    uint32_t has4;    // phase A: a run of >= 4 zeros lies in or ends in this chunk
    uint32_t umin;    // phase A: smallest u of the chunk
    uint32_t zmask;   // phase A: bit i = sample i of the chunk is zero
};

template <class G>
LACX_HD void thread_init(Thread<G>& th, uint32_t n, int tid) {
    th.tid = tid;
    th.a = tid * G::CH;
    th.n = n;
    const int rem = (int)n - th.a;
    th.cnt = rem < 0 ? 0 : (rem > G::CH ? G::CH : rem);
}

// ---------------------------------------------------------------------------------------------
// sample staging: x[sw(j)] = sample j (0 beyond n)
// ---------------------------------------------------------------------------------------------
LACX_HD int32_t slot_combine(int kind, int32_t l, int32_t r) {
    if (kind == CH_L) return l;
    if (kind == CH_R) return r;
    if (kind == CH_M) return (int32_t)((uint32_t)l + (uint32_t)r) >> 1;
    return (int32_t)((uint32_t)l - (uint32_t)r);
}

struct alignas(16) Vec16 {
    uint32_t w[4];
};

template <int NV>
LACX_HD void load_vec16(const void* p, uint32_t* w /* 4*NV */) {
    const Vec16* q = static_cast<const Vec16*>(p);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const Vec16 t = q[k];
        w[4 * k] = t.w[0];
        w[4 * k + 1] = t.w[1];
        w[4 * k + 2] = t.w[2];
        w[4 * k + 3] = t.w[3];
    }
}

// three bytes starting at byte `sh` (0..3) of the little-endian pair (lo, hi)
LACX_HD uint32_t bytes3(uint32_t lo, uint32_t hi, int sh) {
    const uint32_t x = sh == 0 ? lo : sh == 1 ? (lo >> 8) : sh == 2 ? ((lo >> 16) | (hi << 16)) : ((lo >> 24) | (hi << 8));
    return x & 0xFFFFFFu;
}

LACX_HD bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// The CH consecutive frames of one thread are one contiguous 16-byte-aligned span in every supported layout
// when the caller's buffer is 16-byte aligned: they can be fetched with 16-byte loads (a wave then touches each
// cache line once instead of CH times).  span_ok() says whether that holds for a thread's span; stage_span()
// does the fetch and must only be called when it does.  Values are identical to the per-sample path.
template <int CH>
LACX_HD bool span_ok(const SlotSrc& s, int64_t first) {
    if (s.layout == PCM_PLANAR_I32) {
        const bool needa = s.kind != CH_R, needb = s.kind != CH_L;
        return CH % 4 == 0 && (!needa || aligned16(s.a + first)) && (!needb || aligned16(s.b + first));
    }
    const uint8_t* base = reinterpret_cast<const uint8_t*>(s.a);
    if (s.layout == PCM_INTERLEAVED_I16)
        return s.channels == 2 ? (CH % 4 == 0 && aligned16(base + first * 4)) : (CH % 8 == 0 && aligned16(base + first * 2));
    return CH % 16 == 0 && aligned16(base + first * (s.channels == 2 ? 6 : 3));
}

template <int CH>
LACX_HD void stage_span(const SlotSrc& s, int64_t first, int32_t* v) {
    if (s.layout == PCM_PLANAR_I32) {
        constexpr int NV = CH / 4 > 0 ? CH / 4 : 1;
        const bool needa = s.kind != CH_R, needb = s.kind != CH_L;
        uint32_t la[4 * NV] = {}, lb[4 * NV] = {};
        if (needa) load_vec16<NV>(s.a + first, la);
        if (needb) load_vec16<NV>(s.b + first, lb);
#pragma unroll
        for (int i = 0; i < CH; ++i) v[i] = slot_combine(s.kind, (int32_t)la[i], (int32_t)lb[i]);
        return;
    }
    if (s.layout == PCM_INTERLEAVED_I16) {
        if (s.channels == 2) {
            constexpr int NV = CH / 4 > 0 ? CH / 4 : 1;
            uint32_t w[4 * NV];
            load_vec16<NV>(reinterpret_cast<const uint8_t*>(s.a) + first * 4, w);
#pragma unroll
            for (int i = 0; i < CH; ++i)
                v[i] = slot_combine(s.kind, (int32_t)(int16_t)(w[i] & 0xFFFFu), (int32_t)(int16_t)(w[i] >> 16));
            return;
        }
        constexpr int NV = CH / 8 > 0 ? CH / 8 : 1;
        uint32_t w[4 * NV];
        load_vec16<NV>(reinterpret_cast<const uint8_t*>(s.a) + first * 2, w);
#pragma unroll
        for (int i = 0; i < CH; ++i) v[i] = (int32_t)(int16_t)((w[(i / 2) % (4 * NV)] >> (16 * (i & 1))) & 0xFFFFu);
        return;
    }
    // packed 24-bit little-endian
    if (s.channels == 2) {
        constexpr int ND = CH * 6 / 4 >= 4 ? CH * 6 / 4 : 4;
        uint32_t w[ND];
        load_vec16<ND / 4>(reinterpret_cast<const uint8_t*>(s.a) + first * 6, w);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int ol = (i * 6) % (ND * 4), orr = (i * 6 + 3) % (ND * 4);
            v[i] = slot_combine(s.kind, sext24(bytes3(w[ol / 4], w[(ol / 4 + 1 < ND) ? ol / 4 + 1 : ol / 4], ol & 3)),
                                sext24(bytes3(w[orr / 4], w[(orr / 4 + 1 < ND) ? orr / 4 + 1 : orr / 4], orr & 3)));
        }
        return;
    }
    constexpr int ND = CH * 3 / 4 >= 4 ? CH * 3 / 4 : 4;
    uint32_t w[ND];
    load_vec16<ND / 4>(reinterpret_cast<const uint8_t*>(s.a) + first * 3, w);
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int o = (i * 3) % (ND * 4);
        v[i] = sext24(bytes3(w[o / 4], w[(o / 4 + 1 < ND) ? o / 4 + 1 : o / 4], o & 3));
    }
}

// True when the predicate holds in every lane of the wave (a scalar value on the device: branching on it costs
// one scalar branch, whereas a per-lane condition makes the wave walk through both sides piecewise).
LACX_HD bool wave_all(bool pred) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __all(pred) != 0;
#else
    return pred;
#endif
}

// The CH samples [first, first + CH) of a slot into v[]: one span fetch when all of them exist (cnt == CH) and
// the spans of the whole wave are aligned, else sample by sample with the index clamped to `last` (callers mask
// what lies beyond).
template <int CH>
LACX_HD void load_chunk(const SlotSrc& src, int64_t first, int cnt, int64_t last, int32_t* v) {
    if (wave_all(cnt == CH && span_ok<CH>(src, first))) {
        stage_span<CH>(src, first, v);
        return;
    }
    // per-sample path: all loads are issued before the first use
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int64_t idx = first + i;
        v[i] = slot_fetch(src, idx < last ? idx : last);
    }
}

// Returns the OR of the chunk's samples (zero: the chunk is silent).
template <class G, class M>
LACX_HD uint32_t stage_samples(const Thread<G>& th, M& sh, const SlotSrc& src, int64_t start) {
    int32_t v[G::CH];
    load_chunk<G::CH>(src, start + th.a, th.cnt, start + (int64_t)th.n - 1, v);
    int32_t* col = &sh.xp.x[th.tid];
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < G::CH; ++i) {
        const int32_t x = (i < th.cnt) ? v[i] : 0;
        col[i * G::T] = x;
        any |= (uint32_t)x;
    }
    return any;
}

// bit-sliced add of two W-bit counters held as W words (bit b of word l = bit l of the count of plane b)
template <int W>
LACX_HD void sliced_add(const uint32_t* a, const uint32_t* b, uint32_t* out /* W+1 words */) {
    uint32_t carry = 0;
#pragma unroll
    for (int l = 0; l < W; ++l) {
        const uint32_t x = a[l] ^ b[l];
        out[l] = x ^ carry;
        carry = (a[l] & b[l]) | (x & carry);
    }
    out[W] = carry;
}

template <class G>
LACX_HD void plane_counts(const uint32_t* u, uint32_t* cs) {
    if (G::CH == 16) {
        uint32_t l1[8][2], l2[4][3], l3[2][4], l4[5];
#pragma unroll
        for (int i = 0; i < 8; ++i) sliced_add<1>(&u[2 * i], &u[2 * i + 1], l1[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) sliced_add<2>(l1[2 * i], l1[2 * i + 1], l2[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) sliced_add<3>(l2[2 * i], l2[2 * i + 1], l3[i]);
        sliced_add<4>(l3[0], l3[1], l4);
#pragma unroll
        for (int l = 0; l < G::LV; ++l) cs[l] = l4[l];
    } else {
        uint32_t l1[2][2], l2[3];
        sliced_add<1>(&u[0], &u[1], l1[0]);
        sliced_add<1>(&u[2], &u[3], l1[1]);
        sliced_add<2>(l1[0], l1[1], l2);
#pragma unroll
        for (int l = 0; l < G::LV; ++l) cs[l] = l2[l];
    }
}

// ---------------------------------------------------------------------------------------------
// Phase R: residual of candidate `cand` over the chunk -> u[] in LDS (plain, no flags), chunk sum,
// last non-zero index, bit-sliced plane counts.
// cand 0..4 fixed orders (ref block/encoder.cpp:265-295), 5 FIR {3,-1}>>2 (:297-309), 6..10 LPC orders
// 4..12 with open-loop Q15 prediction and zero history before the slot (ref lpc/lpc.cpp:38-61).
// Fixed differences are taken in wrapping 32-bit arithmetic: the true values fit int32 for
// |x| <= 2^24, so the low 32 bits equal the reference's int64 results.
// ---------------------------------------------------------------------------------------------
// Part 1: the chunk's zigzag residual into registers.
template <class G, class M>
LACX_HD void phase_r_residual(Thread<G>& th, const M& sh, int cand, uint32_t* u /* CH */) {
    // x[a-12 .. a+CH): element `el` of chunk tid+co sits at row el, column tid+co of the transposed
    // image, i.e. at a compile-time offset from one base address (no per-candidate address arithmetic).
    int32_t xh[G::CH + 12];
#pragma unroll
    for (int i = 0; i < G::CH + 12; ++i) {
        const int d = i - 12;
        const int co = (d >= 0) ? d / G::CH : -((-d + G::CH - 1) / G::CH);
        const int el = d - co * G::CH;
        const int tt = th.tid + co;
        // own base per column so that every row is base + a 16-bit immediate offset
        const int32_t* col = &sh.xp.x[tt < 0 ? 0 : tt];
        xh[i] = (tt >= 0) ? col[el * G::T] : 0;
    }
    const int32_t* x = xh + 12;
    if (cand <= 4) {
        // one loop per order (block-uniform branch) rather than all four formulas plus selects per sample
        auto fixed = [&](auto order_tag) {
            constexpr int K = decltype(order_tag)::value;
#pragma unroll
            for (int i = 0; i < G::CH; ++i) {
                const uint32_t x0 = (uint32_t)x[i], x1 = (uint32_t)x[i - 1], x2 = (uint32_t)x[i - 2],
                               x3 = (uint32_t)x[i - 3], x4 = (uint32_t)x[i - 4];
                uint32_t r = x0;
                if (K == 1) r = x0 - x1;
                if (K == 2) r = x0 - 2u * x1 + x2;
                if (K == 3) r = x0 - 3u * x1 + 3u * x2 - x3;
                if (K == 4) r = x0 - 4u * x1 + 6u * x2 - 4u * x3 + x4;
                if (K > 0 && th.a + i < K) r = x0;  // warm-up samples are stored raw
                u[i] = (i < th.cnt) ? zigzag32((int32_t)r) : 0u;
            }
        };
        if (cand == 0) fixed(std::integral_constant<int, 0>{});
        else if (cand == 1) fixed(std::integral_constant<int, 1>{});
        else if (cand == 2) fixed(std::integral_constant<int, 2>{});
        else if (cand == 3) fixed(std::integral_constant<int, 3>{});
        else fixed(std::integral_constant<int, 4>{});
    } else if (cand == 5) {
#pragma unroll
        for (int i = 0; i < G::CH; ++i) {
            const int j = th.a + i;
            const int64_t pred = (3LL * (int64_t)x[i - 1] - (int64_t)x[i - 2]) >> 2;
            const int32_t r = (j >= 2) ? (int32_t)((int64_t)x[i] - pred) : x[i];
            u[i] = (i < th.cnt) ? zigzag32(r) : 0u;
        }
    } else {
        // one loop per tap count (block-uniform branch): an order-4 candidate multiplies 4 taps, not 12
        const int ci = cand - 6;
        const int ord = sh.lpc.used[ci];
        auto lpc = [&](auto taps_tag) {
            constexpr int TAPS = decltype(taps_tag)::value;
            int32_t c[TAPS + 1];
#pragma unroll
            for (int t = 1; t <= TAPS; ++t) c[t] = (t <= ord) ? (int32_t)sh.lpc.coef[ci][t] : 0;
#pragma unroll
            for (int i = 0; i < G::CH; ++i) {
                int64_t acc = 0;
#pragma unroll
                for (int t = 1; t <= TAPS; ++t) acc += (int64_t)c[t] * (int64_t)x[i - t];
                const int32_t r = (int32_t)((int64_t)x[i] - (acc >> 15));
                u[i] = (i < th.cnt) ? zigzag32(r) : 0u;
            }
        };
        if (ord <= 4) lpc(std::integral_constant<int, 4>{});
        else if (ord <= 6) lpc(std::integral_constant<int, 6>{});
        else if (ord <= 8) lpc(std::integral_constant<int, 8>{});
        else if (ord <= 10) lpc(std::integral_constant<int, 10>{});
        else lpc(std::integral_constant<int, 12>{});
    }
}

// Part 2 (only for candidates that survive the pruning): the residual into LDS, the chunk sum, the last non-zero
// index and the bit-sliced plane counts.
template <class G, class M>
LACX_HD void phase_r_store(Thread<G>& th, M& sh, const uint32_t* u) {
    uint64_t s = 0;
    int32_t lastnz = -1;
#pragma unroll
    for (int i = 0; i < G::CH; ++i) {
        s += u[i];
        if (u[i] != 0) lastnz = th.a + i;
        sh.u[i * G::T + th.tid] = u[i];
    }
    sh.tabP[th.tid] = s;
    sh.tabNZ[th.tid] = lastnz;
    plane_counts<G>(u, th.cs);
}

template <class G, class M>
LACX_HD void phase_r(Thread<G>& th, M& sh, int cand) {
    uint32_t u[G::CH];
    phase_r_residual(th, sh, cand, u);
    phase_r_store(th, sh, u);
}

// ---------------------------------------------------------------------------------------------
// Pass 1: the pruning-bound partials of ALL eleven candidates in one walk over the chunk.
// The bound (candidate_lower_bound) needs four block sums per candidate: sum of bit_width(u) + 1, the zeros, the
// fours (u == 4) and the zero runs that end inside a chunk.  None of them needs the zigzag value itself:
//   bit_width(u) = 1 + bit_width(r ^ (r >> 31)) for r != 0, so clz(u | 1) + 1 = lead_m(r) := the number of leading bits
//   of r that equal its sign bit, capped at 32 (ONE instruction, v_ffbh_i32; r = 0 and r = -1 give 32);
//   u == 0 <=> r == 0 and u == 4 <=> r == 2.
// The sample window is loaded once for all candidates, the five fixed orders are nested differences (order K + 1 is
// the first difference of order K: four subtractions per sample for all of them instead of ten multiply-adds), and
// nothing is stored.  Per sample and candidate that leaves lead_m + min + add and a 2-bit code (min + shift-add) from
// which zeros, fours and run ends are counted once per chunk, against the residual + zigzag + clz + three compares and
// seven scalar instructions of a candidate-by-candidate walk with wave ballots (ref block/encoder.cpp:362-407 computes
// every residual in full and costs it exactly).
// Positions beyond the slot count as r = 0 (a zero, no run end); the driver takes them out again.
// ---------------------------------------------------------------------------------------------
LACX_HD uint32_t lead_m(int32_t r) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t f;  // v_ffbh_i32: position of the first bit that differs from the sign bit, -1 when none does
    asm("v_ffbh_i32 %0, %1" : "=v"(f) : "v"(r));
    return f < 32u ? f : 32u;
#else
    const uint32_t v = (uint32_t)(r ^ (r >> 31));
    return v ? (uint32_t)__builtin_clz(v) : 32u;
#endif
}

struct BoundPartials {
    uint32_t msum;  // this thread: sum over the CH positions of lead_m(r)
#if defined(__HIP_DEVICE_COMPILE__)
    // Device: two bits per position, first sample in the highest of the CH pairs: min(uint32(r), 3), i.e. 0 for a zero
    // and 2 for r == 2 (u == 4) -- one min and one shift-add per sample; zeros, fours and run ends are counted from it
    // once per chunk (bound_counts).  (Counting them per sample from wave ballots -- two compares and seven scalar
    // instructions per sample and candidate -- made the scalar unit the bottleneck of this pass.)
    uint32_t code;
#else
    // host simulator: this thread's counts
    uint32_t nz;    // r == 0 (incl. the positions beyond the slot)
    uint32_t n4;    // r == 2, i.e. u == 4
    uint32_t ends;  // a zero followed by a non-zero inside a chunk
    uint32_t zprev;
#endif
};

LACX_HD void bound_init(BoundPartials& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    b.msum = b.code = 0;
#else
    b.msum = b.nz = b.n4 = b.ends = b.zprev = 0;
#endif
}

LACX_HD void bound_add(BoundPartials& b, int32_t r) {
    b.msum += lead_m(r);
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t v = (uint32_t)r;
    b.code = (b.code << 2) + (v < 3u ? v : 3u);
#else
    const uint32_t z = r == 0 ? 1u : 0u;
    b.nz += z;
    b.ends += (b.zprev && !z) ? 1u : 0u;
    b.zprev = z;
    b.n4 += r == 2 ? 1u : 0u;
#endif
}

// This thread's zeros | fours << 11 | run ends << 22 over its CH positions (each field leaves room for the sum over a
// wave: at most 64 * 16 = 1024 zeros or fours and 64 * 8 run ends).
template <int CH>
LACX_HD uint32_t bound_counts(const BoundPartials& b) {
    static_assert(CH <= 16, "two bits per position in one word");
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr uint32_t kLow = CH == 16 ? 0x55555555u : ((1u << (2 * (CH & 15))) - 1u) & 0x55555555u;  // low bit of every pair in use
    const uint32_t lo = b.code & kLow, hi = (b.code >> 1) & kLow;
    const uint32_t zero = ~(lo | hi) & kLow;                  // pair == 0
    const uint32_t four = hi & ~lo;                           // pair == 2
    // pair p = position CH - 1 - p: a zero (pair p) followed by a non-zero (pair p - 1); the last position ends nothing
    const uint32_t ends = zero & ~(zero << 2) & ~1u;
    return (uint32_t)__builtin_popcount(zero) | ((uint32_t)__builtin_popcount(four) << 11) |
           ((uint32_t)__builtin_popcount(ends) << 22);
#else
    return b.nz | (b.n4 << 11) | (b.ends << 22);
#endif
}

// FULL: every position of every chunk lies inside the slot (n == MAXN, block-uniform): no per-sample masking.
// The window of a chunk: its CH samples and the twelve before it (as in phase_r_residual).
template <class G, class M>
LACX_HD void pass1_window(const Thread<G>& th, const M& sh, int32_t* xh /* CH + 12 */) {
#pragma unroll
    for (int i = 0; i < G::CH + 12; ++i) {
        const int d = i - 12;
        const int co = (d >= 0) ? d / G::CH : -((-d + G::CH - 1) / G::CH);
        const int el = d - co * G::CH;
        const int tt = th.tid + co;
        const int32_t* col = &sh.xp.x[tt < 0 ? 0 : tt];
        xh[i] = (tt >= 0) ? col[el * G::T] : 0;
    }
}

// Fixed orders 0..4 and the FIR predictor -> out[0..5].
template <class G, bool FULL, class M>
LACX_HD void pass1_bounds_fixed(const Thread<G>& th, const M& sh, BoundPartials* out /* 6 */) {
    int32_t xh[G::CH + 12];
    pass1_window(th, sh, xh);
    const int32_t* x = xh + 12;
    const bool first = th.a == 0;  // the slot's first chunk: warm-up samples are taken raw (ref block/encoder.cpp:265-309)
    auto live = [&](int i, int32_t r) { return FULL ? r : ((i < th.cnt) ? r : 0); };
    {
        // fixed orders 0..4 and the FIR predictor
        BoundPartials b[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) bound_init(b[c]);
        // the differences that precede the chunk (wrapping 32-bit: the true values fit, see phase_r_residual)
        const uint32_t e1 = (uint32_t)x[-1] - (uint32_t)x[-2], e1b = (uint32_t)x[-2] - (uint32_t)x[-3],
                       e1c = (uint32_t)x[-3] - (uint32_t)x[-4];
        const uint32_t e2 = e1 - e1b, e2b = e1b - e1c;
        uint32_t p1 = e1, p2 = e2, p3 = e2 - e2b;
#pragma unroll
        for (int i = 0; i < G::CH; ++i) {
            const uint32_t x0 = (uint32_t)x[i];
            const uint32_t d1 = x0 - (uint32_t)x[i - 1], d2 = d1 - p1, d3 = d2 - p2, d4 = d3 - p3;
            p1 = d1;
            p2 = d2;
            p3 = d3;
            const int32_t fir = (int32_t)(x0 - (uint32_t)((3 * x[i - 1] - x[i - 2]) >> 2));  // |3 x1 - x2| <= 2^26
            const bool w1 = first && i < 1, w2 = first && i < 2, w3 = first && i < 3, w4 = first && i < 4;
            bound_add(b[0], live(i, (int32_t)x0));
            bound_add(b[1], live(i, (int32_t)(w1 ? x0 : d1)));
            bound_add(b[2], live(i, (int32_t)(w2 ? x0 : d2)));
            bound_add(b[3], live(i, (int32_t)(w3 ? x0 : d3)));
            bound_add(b[4], live(i, (int32_t)(w4 ? x0 : d4)));
            bound_add(b[5], live(i, w2 ? (int32_t)x0 : fir));
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) out[c] = b[c];
    }
}

// The LPC candidates (orders 4, 6, 8, 10, 12: open-loop Q15, ref lpc/lpc.cpp:38-61) -> out[0..4].
// lpc_off: the LPC candidates are switched off (diagnostic ablation).  out[c] of an unavailable LPC candidate is zeroed.
template <class G, bool FULL, class M>
LACX_HD void pass1_bounds_lpc(const Thread<G>& th, const M& sh, bool lpc_off, BoundPartials* out /* 5 */) {
    int32_t xh[G::CH + 12];
    pass1_window(th, sh, xh);
    const int32_t* x = xh + 12;
    auto live = [&](int i, int32_t r) { return FULL ? r : ((i < th.cnt) ? r : 0); };
    for (int ci = 0; ci < 5; ++ci) {
        BoundPartials b;
        bound_init(b);
        const int ord = lpc_off ? 0 : (int)sh.lpc.used[ci];
        auto lpc = [&](auto taps_tag) {
            constexpr int TAPS = decltype(taps_tag)::value;
            int32_t c[TAPS + 1];
#pragma unroll
            for (int t = 1; t <= TAPS; ++t) c[t] = (t <= ord) ? (int32_t)sh.lpc.coef[ci][t] : 0;
#pragma unroll
            for (int i = 0; i < G::CH; ++i) {
                int64_t acc = 0;
#pragma unroll
                for (int t = 1; t <= TAPS; ++t) acc += (int64_t)c[t] * (int64_t)x[i - t];
                // only the low 32 bits of the prediction matter (the residual fits int32 in the validated domain)
                const int32_t r = (int32_t)((uint32_t)x[i] - (uint32_t)((uint64_t)acc >> 15));
                bound_add(b, live(i, r));
                }
        };
        if (ord == 0) {
            // not available (uniform): nothing to add
        } else if (ord <= 4) lpc(std::integral_constant<int, 4>{});
        else if (ord <= 6) lpc(std::integral_constant<int, 6>{});
        else if (ord <= 8) lpc(std::integral_constant<int, 8>{});
        else if (ord <= 10) lpc(std::integral_constant<int, 10>{});
        else lpc(std::integral_constant<int, 12>{});
        // (selects, not out[ci]: an array indexed by the loop counter lives in scratch memory)
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            if (c == ci) out[c] = b;
        }
    }
}

// all eleven (the host simulator's form)
template <class G, bool FULL, class M>
LACX_HD void pass1_bounds(const Thread<G>& th, const M& sh, bool lpc_off, BoundPartials* out /* 11 */) {
    pass1_bounds_fixed<G, FULL>(th, sh, out);
    pass1_bounds_lpc<G, FULL>(th, sh, lpc_off, out + 6);
}

// Horner from plane counts C[0..29] to A[k] = sum_j (u_j >> k), k = 0..kmax.
LACX_HD void planes_to_ksums(const uint32_t* C, uint64_t* A, int kmax) {
    uint64_t acc = 0;
#pragma unroll
    for (int b = 29; b >= 0; --b) {
        acc = (acc << 1) + C[b];
        if (b <= 15) A[b] = (b <= kmax) ? acc : 0;
    }
}

// estimate_initial_k (ref block/encoder.cpp:121-158): argmin over k=0..12 of sum (u>>k) + m*(1+k).
LACX_HD uint32_t pick_initial_k(const uint64_t* A, uint32_t m) {
    uint32_t best_k = 0;
    uint64_t best = ~0ull;
#pragma unroll
    for (uint32_t k = 0; k <= 12; ++k) {
        const uint64_t c = A[k] + (uint64_t)m * (1u + k);
        if (c < best) {
            best = c;
            best_k = k;
        }
    }
    return best_k;
}

// estimate_static_k + estimate_static_rice_bits (ref block/encoder.cpp:160-188).
LACX_HD uint32_t pick_static_k(const uint64_t* A, uint32_t m, uint64_t* bits) {
    uint32_t best_k = 0;
    uint64_t best = ~0ull;
#pragma unroll
    for (uint32_t k = 0; k <= 15; ++k) {
        const uint64_t c = A[k] + (uint64_t)m * (1u + k);
        if (c < best) {
            best = c;
            best_k = k;
        }
    }
    *bits = best;
    return best_k;
}

// Phase A: unbiased k after every sample -> the micro-window flags of every sample (ref rice.hpp:68-80) as two 16-bit
// masks per chunk (tabZM: quotient == 0, tabLM: quotient > 3) and the chunk's packed flag counts.  u itself is not
// touched (round 4: the flags used to go into bits 30 / 31 of u -- sixteen LDS writes per thread and candidate, a mask
// on every later read and a pass over the whole residual to strip them before the partition search).
template <class G, bool NARROW, class M>
LACX_HD void phase_a(Thread<G>& th, M& sh) {
    uint64_t P = sh.tabP[th.tid];
    uint32_t cnt = 0;
    uint32_t c = (uint32_t)th.a;
    uint32_t zm = 0;  // bit i: sample i of the chunk is zero
    uint32_t mn = 0xFFFFFFFFu;
    uint32_t fzm = 0;  // bit i: sample i has the zero-quotient flag
    uint32_t flm = 0;  // bit i: sample i has the large-quotient flag
    for (int i = 0; i < th.cnt; ++i) {
        const uint32_t u = sh.u[i * G::T + th.tid];
        mn = u < mn ? u : mn;
        P += u;
        ++c;
        const uint32_t km = kmean_t<NARROW>(P, c);
        const uint32_t q = u >> km;  // km <= 31 and u < 2^30
        const uint32_t fl = q > 3u, fz = q == 0u;
        cnt += fl + (fz << 16);
        zm |= (u == 0u ? 1u : 0u) << i;
        fzm |= fz << i;
        flm |= fl << i;
    }
    sh.tabF[th.tid] = cnt;
    sh.tabZM[th.tid] = (uint16_t)fzm;
    sh.tabLM[th.tid] = (uint16_t)flm;
    if constexpr (requires { sh.tabUZ[0]; }) sh.tabUZ[th.tid] = (uint16_t)zm;
    th.umin = mn;
    th.zmask = zm;
    // Zero-run mode can only matter when some run of >= 4 zeros exists (ref block/encoder.cpp:224-247 sets
    // has_run only then).  Such a run lies inside one chunk, or crosses into a chunk: then the zeros ending
    // just before the chunk plus the chunk's leading zeros reach 4.
    const uint32_t lead = (uint32_t)ctz32(~zm);
    const int32_t before = th.a - 1 - sh.tabNZ[th.tid];
    th.has4 = (th.cnt > 0 && ((zm & (zm >> 1) & (zm >> 2) & (zm >> 3)) != 0u || (uint32_t)before + lead >= 4u)) ? 1u : 0u;
}

// Sample j as seen by the zero-run lookahead: its u, or 1 ("not a zero") at/after `limit`.
// j may run up to 3 past the slot; the read then lands in the pad words at the end of u[] and is discarded.
template <class G, class M>
LACX_HD uint32_t peek_u(const M& sh, uint32_t j, uint32_t limit) {
    const uint32_t idx = (j % (uint32_t)G::CH) * (uint32_t)G::T + (j / (uint32_t)G::CH);
    const uint32_t v = sh.u[idx] & 0x3FFFFFFFu;
    return (j < limit) ? v : 1u;
}

// Packed flag counts (q > 3 in bits 0..15, q == 0 in bits 16..31) over the 96 samples that precede chunk t:
// the sum of the W96 chunk counts before it (fewer at the start of the slot).
template <class G>
LACX_HD uint32_t window_flags(const uint32_t* tabF, int t) {
    uint32_t d = 0;
#pragma unroll
    for (int k = 1; k <= G::W96; ++k) d += (t >= k) ? tabF[t - k] : 0u;
    return d;
}

// Phase B (stateful, whole block as one segment): rice/bin/zero-run bit costs
// (ref block/encoder.cpp:201-263 with Rice::adapt_k, rice.hpp:45-114) of samples [i0, i1) of chunk t.
// The parameter in force for a sample is a function of running sums only (prefix sum, count, the sum of the last 256,
// the flag counts of the last 96), never of the parameter before it: a span can start anywhere in the chunk once those
// sums have been carried over the samples it skips (a few adds each).  (Sharing the walk of a chunk out over four lanes
// that way was measured and lost to walking whole chunks: the carry-over and the set-up outweigh the shorter chain.)
// TRIPS > 0: i1 - i0 == TRIPS for every lane (a fixed trip count, no per-lane loop exit).
// STEADY: the chunk starts at sample 1024 or later (every wave but the first of a 1024-thread slot): the window taps
// exist and the count thresholds of the bias are met throughout.
struct ChunkCosts {
    unsigned long long rice, bin, zr;
    uint32_t hasrun;
};

template <class G, bool NARROW, bool ZR = true, int TRIPS = 0, bool STEADY = false>
LACX_HD ChunkCosts phase_b_span(const Smem<G>& sh, uint32_t n, int t, int i0, int i1, uint32_t k0) {
    const int a = t * G::CH;
    uint64_t P = sh.tabP[t];                                   // P_{a-1}
    uint64_t W = (STEADY || t >= G::W256) ? sh.tabP[t - (STEADY || t >= G::W256 ? G::W256 : 0)] : 0;    // P_{a-1-256}
    // packed flag counts over the 96 samples before the chunk = its W96 predecessors' chunk counts (the window
    // starts on a chunk boundary); kept up to date sample by sample below
    uint32_t D = window_flags<G>(sh.tabF, t);
    const uint32_t m256 = (STEADY || t >= G::W256) ? 0xFFFFFFFFu : 0u;   // window taps exist from chunk W256 / W96 on
    const int t256 = (STEADY || t >= G::W256) ? t - G::W256 : t;
    const int t96 = (STEADY || t >= G::W96) ? t - G::W96 : t;
    // micro-window flags of this chunk's samples (they enter the 96-window) and of the chunk 96 samples back (they leave
    // it): large-quotient flag of sample i in bit i, zero-quotient flag in bit 16 + i -- the packed counts' layout
    const uint32_t fin = (uint32_t)sh.tabLM[t] | ((uint32_t)sh.tabZM[t] << 16);
    const uint32_t fout = (STEADY || t >= G::W96) ? ((uint32_t)sh.tabLM[t96] | ((uint32_t)sh.tabZM[t96] << 16)) : 0u;
    uint32_t c = (uint32_t)a;
    int32_t f = a - 1 - sh.tabNZ[t];  // zeros ending just before the chunk
    // carry the sums over the samples before the span (none when the span opens the chunk)
    for (int j = 0; j < i0; ++j) {
        const uint32_t u = sh.u[j * G::T + t];
        P += u;
        ++c;
        W += sh.u[j * G::T + t256] & m256;
        D += (fin >> j) & 0x00010001u;
        D -= (fout >> j) & 0x00010001u;
        f = (f + 1) & (int32_t)(0u - flag01(u == 0u));
    }
    // k in force for the first sample of the span: the value returned after the sample before it
    uint32_t kin = k0;
    if (STEADY || (c > 0u && i1 > i0)) kin = biased_k<NARROW, STEADY>(kmean_t<NARROW>(P, c), P, W, D, c);
    // span sums: 32 bits suffice on the narrow path (each cost <= u + 36 and the block's sum of u is below kNarrowLimit)
    using Acc = typename std::conditional<NARROW, uint32_t, unsigned long long>::type;
    Acc rice = 0, bin = 0, zr = 0;
    uint32_t hasrun = 0;
    const int ifirst = i0 & (G::CH - 1);
    uint32_t w0 = sh.u[ifirst * G::T + t];  // own sample incl. flags
    uint32_t n1 = 1, n2 = 1, n3 = 1;
    if (ZR) {
        n1 = peek_u<G>(sh, (uint32_t)(a + i0) + 1u, n);
        n2 = peek_u<G>(sh, (uint32_t)(a + i0) + 2u, n);
        n3 = peek_u<G>(sh, (uint32_t)(a + i0) + 3u, n);
    }
    // the window tap of the current trip is fetched one trip ahead (its latency hides behind the trip's arithmetic)
    uint32_t tap256 = sh.u[ifirst * G::T + t256];
    const int last = TRIPS > 0 ? i0 + TRIPS : i1;
#pragma nounroll  // one sample per trip: the 16-fold body does not fit the register budget
    for (int i = i0; i < last; ++i) {
        const uint32_t cur256 = tap256;
        const int inext = (i + 1) & (G::CH - 1);
        const uint32_t wnext = sh.u[inext * G::T + t];
        tap256 = sh.u[inext * G::T + t256];
        const uint32_t u = w0;
        const uint32_t rc = (u >> kin) + 1u + kin;  // kin <= 31 (biased_k clamps) and u < 2^30: no k >= 31 special case
        rice += rc;
        bin += 2u + ((u <= 4u) ? (u < 1u ? u : 1u) : rc);  // 2 for a zero, 3 for 1..4, else 2 + the Rice code
        if (ZR) {
            const uint32_t z = flag01(u == 0u);
            f = (f + 1) & (int32_t)(0u - z);
            const uint32_t ahead = (n1 != 0) ? 0u : ((n2 != 0) ? 1u : ((n3 != 0) ? 2u : 3u));
            const uint32_t in4 = z & flag01((uint32_t)f + ahead >= 4u);
            const uint32_t esc = 1u << ((kin + 3u) < 24u ? (kin + 3u) : 24u);
            const uint32_t plain = 2u + ((u > esc) ? 32u : rc);              // not inside a run of >= 4
            const uint32_t token = 5u + (((uint32_t)(f - 4)) >> 2);          // last sample of such a run
            const uint32_t runend = in4 & flag01(n1 != 0);
            zr += (plain & (in4 - 1u)) | (token & (0u - runend));            // in4 - 1: all ones outside a run
            hasrun |= runend;
        }
        // state after this sample -> k for the next one
        P += u;
        ++c;
        W += cur256 & m256;
        D += (fin >> i) & 0x00010001u;    // sample j enters the window ...
        D -= (fout >> i) & 0x00010001u;   // ... sample j-96 leaves it
        kin = biased_k<NARROW, STEADY>(kmean_t<NARROW>(P, c), P, W, D, c);
        // slide the lookahead window
        w0 = wnext;
        if (ZR) {
            n1 = n2;
            n2 = n3;
            n3 = peek_u<G>(sh, (uint32_t)(a + i) + 4u, n);
        }
    }
    ChunkCosts out;
    out.rice = rice;
    out.bin = bin;
    out.zr = zr;
    out.hasrun = hasrun;
    return out;
}

// The whole chunk of the thread.  FULL: every chunk of the slot is complete (n == MAXN, block-uniform).
template <class G, bool NARROW, bool ZR = true, bool FULL = false, bool STEADY = false>
LACX_HD void phase_b(Thread<G>& th, const Smem<G>& sh, uint32_t k0) {
    const ChunkCosts cc = phase_b_span<G, NARROW, ZR, FULL ? G::CH : 0, STEADY>(sh, th.n, th.tid, 0, th.cnt, k0);
    th.crice = cc.rice;
    th.cbin = cc.bin;
    th.czr = cc.zr;
    th.chasrun = cc.hasrun;
}

// Picks the instance of phase_b: narrow / zr / full are block-uniform; the chunk-position instance (STEADY) is taken by
// every wave but the first of a multi-wave slot (a >= 64 * CH >= 256, wave-uniform).
template <class G>
LACX_HD void phase_b_dispatch(Thread<G>& th, const Smem<G>& sh, uint32_t k0, bool narrow, bool zr, bool full, bool steady) {
    auto go = [&](auto narrow_t, auto zr_t) {
        constexpr bool N = decltype(narrow_t)::value, Z = decltype(zr_t)::value;
        if (full) {
            if (steady) phase_b<G, N, Z, true, true>(th, sh, k0); else phase_b<G, N, Z, true, false>(th, sh, k0);
        } else {
            if (steady) phase_b<G, N, Z, false, true>(th, sh, k0); else phase_b<G, N, Z, false, false>(th, sh, k0);
        }
    };
    if (narrow) {
        if (zr) go(std::true_type{}, std::true_type{}); else go(std::true_type{}, std::false_type{});
    } else {
        if (zr) go(std::false_type{}, std::true_type{}); else go(std::false_type{}, std::false_type{});
    }
}
template <class G>
LACX_HD void phase_b_dispatch(Thread<G>& th, const Smem<G>& sh, uint32_t k0, bool narrow, bool zr, bool full) {
    phase_b_dispatch<G>(th, sh, k0, narrow, zr, full, G::T > 64 && th.tid >= 64 && 64 * G::CH >= 256);
}

#ifndef LACX_QUEUED_STEADY
#define LACX_QUEUED_STEADY 1
#endif
// The walk of chunk t (t >= 64: the steady instance) by whichever lane the queue of phase B hands it to; the chunk's
// costs are added to the lane's own.
template <class G>
LACX_HD void phase_b_queued(Thread<G>& th, const Smem<G>& sh, int t, uint32_t k0, bool narrow, bool zr, bool full) {
    Thread<G> c;
    thread_init(c, th.n, t);
    phase_b_dispatch<G>(c, sh, k0, narrow, zr, full, LACX_QUEUED_STEADY != 0);
    th.crice += c.crice;
    th.cbin += c.cbin;
    th.czr += c.czr;
    th.chasrun |= c.chasrun;
}

LACX_HD uint32_t census_code(uint32_t u);
LACX_HD uint32_t census_shift_sum(uint32_t cen, uint32_t K);

// Zero-run structure of one chunk from its zero mask (bits 0..CH-1; bits CH..CH+2 = the three samples after it) and the
// number of zeros ending just before it: how many of its samples lie inside a run of >= 4 zeros (they cost nothing until
// the run's last one pays the token, ref block/encoder.cpp:224-247), the tokens of the runs that end in the chunk,
// whether one does, and whether the chunk's first sample is inside such a run.
struct RunShape {
    uint32_t nin4, tokens, hasrun, first_in4;
};
template <int CH>
LACX_HD RunShape run_shape(uint32_t E, int32_t f) {
    RunShape r{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const uint32_t z = (E >> i) & 1u;
        f = (f + 1) & (int32_t)(0u - z);
        const uint32_t after = E >> (i + 1);
        const uint32_t ahead = (after & 1u) == 0u ? 0u : ((after & 2u) == 0u ? 1u : ((after & 4u) == 0u ? 2u : 3u));
        const uint32_t in4 = z & flag01((uint32_t)f + ahead >= 4u);
        const uint32_t runend = in4 & ((after & 1u) ^ 1u);
        r.nin4 += in4;
        r.tokens += (5u + (((uint32_t)(f - 4)) >> 2)) & (0u - runend);
        r.hasrun |= runend;
        if (i == 0) r.first_in4 = in4;
    }
    return r;
}

// Phase B of one chunk without the walk, where that is provably the same thing.  The Rice parameter in force for a sample
// is a function of four running quantities (ref rice.hpp:68-113): the prefix sum P and count c (unbiased k), the sum of
// the last 256 (drift) and the flag counts of the last 96 (micro window).  Over the CH samples of a chunk each of them
// moves inside an interval known from the tables the block scans have already built -- P in [tabP[t], tabP[t+1]], the
// sum leaving the 256-window in chunk t - W256's sum, the flags entering / leaving the 96-window in tabF[t] /
// tabF[t - W96] (the zero flags sample by sample, from tabZM) -- and every comparison of adapt_k is monotone in them.  When all of them come out the same at both ends
// of the intervals the parameter is one constant k for the whole chunk, and the three adaptive costs follow from the
// chunk's k-sum (its plane counts), a census of its small values (bin) and its zero mask (zero-run):
//   rice = sum (u >> k) + CH (1 + k);  bin and zero-run = rice + 2 CH when no sample is <= 4.
// A sample above the zero-run escape sends the chunk to the walk as well.
// Returns false (costs untouched) when any of this cannot be shown; the caller then walks the chunk (phase_b).
// Needs the full windows behind the chunk (every wave but the first of a whole-block slot) and a complete chunk.
template <class G, bool NARROW, bool ZR>
LACX_HD bool phase_b_quick(Thread<G>& th, const Smem<G>& sh) {
    if constexpr (G::T <= 64 || 64 * G::CH <= 256) {
        return false;
    } else {
        const int t = th.tid;
        if (t < 64 || th.cnt != G::CH) return false;  // c > 256 throughout (the drift is armed), all window taps exist
        using Sum = typename std::conditional<NARROW, uint32_t, uint64_t>::type;
        const Sum P0 = (Sum)sh.tabP[t], P1 = (Sum)sh.tabP[t + 1];
        const Sum W0 = (Sum)sh.tabP[t - G::W256], W1 = (Sum)sh.tabP[t - G::W256 + 1];
        const uint32_t c0 = (uint32_t)th.a, c1 = c0 + (uint32_t)G::CH;
        // unbiased k: grows with P, falls with c
        const uint32_t km = kmean_t<NARROW>(P0, c1);
        uint32_t ok = flag01(km == kmean_t<NARROW>(P1, c0));
        // drift (ref rice.hpp:85-95)
        const uint32_t Llo = (uint32_t)(((P0 - W1) + 128u) >> 8), Lhi = (uint32_t)(((P1 - W0) + 128u) >> 8);
        const uint32_t Ulo = (3u * Llo + 3u) >> 2, Uhi = (3u * Lhi + 3u) >> 2;
        const uint32_t Dlo = Llo + (Llo + 3u) / 3u + 1u, Dhi = Lhi + (Lhi + 3u) / 3u + 1u;
        const uint64_t Xlo = NARROW ? (uint64_t)((uint32_t)P0 + (c0 >> 1)) : (uint64_t)P0 + (c0 >> 1);
        const uint64_t Xhi = NARROW ? (uint64_t)((uint32_t)P1 + (c1 >> 1)) : (uint64_t)P1 + (c1 >> 1);
        const uint32_t act_never = flag01(Xhi < (uint64_t)c0), act_always = flag01(Xlo >= (uint64_t)c1);
        const uint32_t up_always = flag01(Xhi < (uint64_t)Ulo * c0), up_never = flag01(Xlo >= (uint64_t)Uhi * c1);
        const uint32_t dn_always = flag01(Xlo >= (uint64_t)Dhi * c1), dn_never = flag01(Xhi < (uint64_t)Dlo * c0);
        ok &= act_never | (act_always & (up_always | up_never) & (dn_always | dn_never));
        const int32_t drift = (int32_t)((act_never ^ 1u) & up_always) - (int32_t)((act_never ^ 1u) & dn_always);
        // micro window (ref rice.hpp:97-105): the counts over the 96 before the chunk, what enters and what leaves
        const uint32_t d0 = window_flags<G>(sh.tabF, t), din = sh.tabF[t], dout = sh.tabF[t - G::W96];
        const int32_t l0 = (int32_t)(d0 & 0xFFFFu), z0 = (int32_t)(d0 >> 16);
        const uint32_t big_always = flag01(l0 - (int32_t)(dout & 0xFFFFu) >= 72), big_never = flag01(l0 + (int32_t)(din & 0xFFFFu) < 72);
        // The zero count sits near its threshold on ordinary material (the flag marks u < 2^k, about four samples in
        // five), so the interval [z0 - leaving, z0 + entering] straddles 77 more often than not: walk the count
        // itself over the chunk's states from the flag masks of the two chunks involved.
        uint32_t sml_always, sml_never;
        {
            const uint32_t zin = sh.tabZM[t], zout = sh.tabZM[t - G::W96];
            int32_t z = z0, zmin = z0, zmax = z0;
#pragma unroll
            for (int i = 0; i + 1 < G::CH; ++i) {
                z += (int32_t)((zin >> i) & 1u) - (int32_t)((zout >> i) & 1u);
                zmin = z < zmin ? z : zmin;
                zmax = z > zmax ? z : zmax;
            }
            sml_always = flag01(zmin >= 77);
            sml_never = flag01(zmax < 77);
        }
        ok &= big_always | (big_never & (sml_always | sml_never));
        int32_t bias = drift + (int32_t)big_always - (int32_t)((big_always ^ 1u) & sml_always);
        bias = bias < -1 ? -1 : (bias > 1 ? 1 : bias);
        int bk = (int)km + bias;
        bk = bk < 0 ? 0 : (bk > 31 ? 31 : bk);
        const uint32_t k = (uint32_t)bk;
        // the values themselves
        uint32_t any = 0;
        Sum ksum = 0;  // (32 bits on the narrow path: the block's whole sum of u is below kNarrowLimit)
#pragma unroll
        for (int l = 0; l < G::LV; ++l) {
            any |= th.cs[l];
            ksum += (Sum)(th.cs[l] >> k) << l;
        }
        if (ZR) ok &= flag01((any >> ((k + 3u) < 24u ? (k + 3u) : 24u)) == 0u);  // nothing beyond the zero-run escape
        if (ok == 0u) return false;
        const Sum rice = ksum + (uint32_t)G::CH * (1u + k);
        // bin: 2 for a zero, 3 for 1..4, else 2 + the Rice code -- the exceptions from a census of the small values
        uint32_t n14 = 0, nbig = (uint32_t)G::CH, small_shifted = 0;
        if (th.umin <= 4u) {
            uint32_t cen = 0;
#pragma unroll
            for (int i = 0; i < G::CH; ++i) cen += census_code(sh.u[i * G::T + t] & 0x3FFFFFFFu);
            n14 = ((cen >> 5) & 31u) + ((cen >> 10) & 31u) + ((cen >> 15) & 31u) + ((cen >> 20) & 31u);
            nbig = (cen >> 25) & 31u;
            small_shifted = census_shift_sum(cen, k);
        }
        th.crice = rice;
        th.cbin = (Sum)(2u * (uint32_t)G::CH + n14 + nbig * (1u + k)) + (ksum - small_shifted);
        // zero-run: a sample inside a run of >= 4 zeros costs nothing until the run's last one pays the token; every
        // other sample 2 + its Rice code.  Which samples those are follows from the chunk's zero mask, the zeros ending
        // just before it and the three samples after it -- no sample is read.
        RunShape rs{0u, 0u, 0u, 0u};
        if (ZR && th.zmask != 0u)
            rs = run_shape<G::CH>(th.zmask | (((uint32_t)sh.tabUZ[t + 1] & 7u) << G::CH), th.a - 1 - sh.tabNZ[t]);
        const uint32_t nin4 = rs.nin4, tokens = rs.tokens, hasrun = rs.hasrun;
        th.czr = ksum + (Sum)(((uint32_t)G::CH - nin4) * (3u + k) + tokens);
        th.chasrun = hasrun;
        return true;
    }
}

template <class G>
LACX_HD bool phase_b_quick_dispatch(Thread<G>& th, const Smem<G>& sh, bool narrow, bool zr) {
    if (narrow) return zr ? phase_b_quick<G, true, true>(th, sh) : phase_b_quick<G, true, false>(th, sh);
    return zr ? phase_b_quick<G, false, true>(th, sh) : phase_b_quick<G, false, false>(th, sh);
}

// Exact lower bound on min(rice, static, zero-run, bin) of a candidate, from four block sums:
//   any Rice code of u costs >= bit_width(u) + 1 bits whatever k is (k = bit_width(u)-1 or bit_width(u) attain it);
//   bin costs the same or more except u == 4 (3 bits against 4);
//   zero-run costs >= 2 + that for every non-zero sample; a maximal run of L zeros costs 3 L (or more) when L < 4 and
//   4 + L / 4 (rounded down) as one token when L >= 4 (ref block/encoder.cpp:224-247).  With Z zeros in R >= max(E, 1)
//   runs (E = run ends seen inside the threads' chunks) the cheapest arrangement is R - 1 single zeros and one long
//   run: the zeros cost >= min(3 Z, 3 (R - 1) + 4 + (Z - R + 1) / 4).  An all-zero block meets the bound exactly.
// A candidate whose bound is >= the best exact cost so far cannot win (the reference replaces the best only
// on a strictly smaller cost, ref block/encoder.cpp:352-359), so its adaptive cost passes can be skipped.
LACX_HD uint64_t candidate_lower_bound(uint32_t g_sum, uint32_t aux_sum, uint32_t ends, uint32_t n, int zero_run) {
    const uint32_t nzero = aux_sum & 0xFFFFu, n4 = aux_sum >> 16;
    const uint64_t lb_rice = g_sum;
    const uint64_t lb_bin = (uint64_t)g_sum - n4;
    uint64_t lb = lb_rice < lb_bin ? lb_rice : lb_bin;
    if (zero_run) {
        uint64_t zeros = 0;
        if (nzero != 0) {
            const uint32_t runs = ends > 1u ? ends : 1u;  // ends <= nzero
            const uint64_t shorts = 3ull * nzero, one_long = 3ull * (runs - 1u) + 4u + ((nzero - runs + 1u) >> 2);
            zeros = shorts < one_long ? shorts : one_long;
        }
        const uint64_t lb_zr = (uint64_t)g_sum - nzero + 2ull * (n - nzero) + zeros;
        if (lb_zr < lb) lb = lb_zr;
    }
    return lb;
}

// Evaluation order of the 11 predictor candidates.  The reference walks them 0..10 and keeps the first
// strictly smallest cost (ref block/encoder.cpp:352-359), i.e. the minimum of (cost, index); any order gives
// the same winner when ties go to the lower index.  Likely winners go first so that the pruning bound bites
// early: fixed order 2, then LPC 12 down to 4, then fixed 1, fixed 3, FIR, fixed 0, fixed 4.
LACX_HD int candidate_at(int i) { return (int)((0x405316789A2ull >> (4 * i)) & 15ull); }

// True when a candidate with lower bound `lb` cannot replace the best (cost, index) seen so far.
LACX_HD bool candidate_pruned(uint64_t lb, int cand, uint64_t best_bits, int best_cand) {
    return best_cand >= 0 && (lb > best_bits || (lb == best_bits && cand > best_cand));
}

// Candidate scoring by thread 0 once the block reductions are in shared memory
// (ref block/encoder.cpp:337-359).  Returns nothing; updates sh.best_*.
template <class G>
LACX_HD void score_candidate(Smem<G>& sh, int cand, uint32_t n, int zero_run, uint32_t k0,
                             const uint32_t* planeTot, const unsigned long long* acc) {
    uint64_t A[16];
    planes_to_ksums(planeTot, A, 15);
    uint64_t sbits;
    const uint32_t sk = pick_static_k(A, n, &sbits);
    const uint64_t rice = acc[0], bin = acc[1];
    const uint32_t hasrun = acc[3] != 0;
    const uint64_t zr = (zero_run && hasrun) ? acc[2] : rice;
    uint64_t a = rice < sbits ? rice : sbits;
    uint64_t b = zr < bin ? zr : bin;
    const uint64_t best = a < b ? a : b;
    if (sh.best_cand < 0 || best < sh.best_bits || (best == sh.best_bits && cand < sh.best_cand)) {
        sh.best_cand = cand;
        sh.best_bits = best;
        sh.best_rice = rice;
        sh.best_zr = zr;
        sh.best_bin = bin;
        sh.best_static = sbits;
        sh.best_k0 = k0;
        sh.best_sk = sk;
        sh.best_hasrun = hasrun;
    }
}

LACX_HD uint32_t initial_k_from_planes(const uint32_t* planes256, uint32_t n) {
    uint64_t A[16];
    planes_to_ksums(planes256, A, 12);
    return pick_initial_k(A, n < 256u ? n : 256u);
}

// ---------------------------------------------------------------------------------------------
// partition search (ref block/encoder.cpp:486-552) on the winning residual
// ---------------------------------------------------------------------------------------------
LACX_HD int max_partition_order(uint32_t n) {  // ref block/encoder.cpp:93-101
    int mp = 0;
    for (int p = 1; p <= kMaxPartitionOrder; ++p) {
        if ((n >> p) < (uint32_t)kMinPartition) break;
        mp = p;
    }
    return mp;
}

// Per-thread packed plane counts (two planes per word: plane w in bits 0..15, plane w+15 in 16..31).
template <class G>
LACX_HD void packed_planes(const Thread<G>& th, uint32_t* words /* 15 */) {
#pragma unroll
    for (int w = 0; w < 15; ++w) {
        uint32_t v = 0;
#pragma unroll
        for (int l = 0; l < G::LV; ++l) {
            v += (((th.cs[l] >> w) & 1u) << l) + (((th.cs[l] >> (w + 15)) & 1u) << (16 + l));
        }
        words[w] = v;
    }
}

// sum_j (u_j >> k) over [s, e), k = 0..15, from the group prefix table + direct sums at ragged ends.
template <class G>
LACX_HD void range_ksums(const Smem<G>& sh, uint32_t s, uint32_t e, uint64_t* A) {
#pragma unroll
    for (int k = 0; k < 16; ++k) A[k] = 0;
    const uint32_t gs = (s + 63u) >> 6, ge = e >> 6;
    uint32_t e0 = e, s1 = e;  // direct ranges [s, e0) and [s1, e)
    if (gs < ge) {
        uint32_t C[30];
#pragma unroll
        for (int w = 0; w < 15; ++w) {
            const uint32_t d = sh.xp.part.grp[w][ge] - sh.xp.part.grp[w][gs];
            C[w] = d & 0xFFFFu;
            C[w + 15] = d >> 16;
        }
        planes_to_ksums(C, A, 15);
        e0 = gs << 6;
        s1 = ge << 6;
    }
    for (uint32_t j = s; j < e0; ++j) {
        const uint32_t u = sh.u[sw<G>((int)j)];
#pragma unroll
        for (int k = 0; k < 16; ++k) A[k] += u >> k;
    }
    for (uint32_t j = s1; j < e; ++j) {
        const uint32_t u = sh.u[sw<G>((int)j)];
#pragma unroll
        for (int k = 0; k < 16; ++k) A[k] += u >> k;
    }
}

LACX_HD void seg_bounds(uint32_t n, int p, uint32_t part, uint32_t* s, uint32_t* e) {
    const uint32_t base = n >> p, parts = 1u << p;  // ref block/encoder.cpp:103-119
    *s = part * base;
    *e = (part + 1u == parts) ? n : (*s + base);
}

// One thread evaluates the static and initial k of one segment.
template <class G>
LACX_HD void seg_static_eval(Smem<G>& sh, uint32_t n, int p, uint32_t part) {
    uint32_t s, e;
    seg_bounds(n, p, part, &s, &e);
    const uint32_t len = e - s;
    uint64_t A[16];
    range_ksums(sh, s, e, A);
    SegInfo si;
    si.sk = (uint8_t)pick_static_k(A, len, &si.sbits);
    const uint32_t m = len < 256u ? len : 256u;
    if (m != len) range_ksums(sh, s, s + m, A);
    si.ak = (uint8_t)pick_initial_k(A, m);
    for (int i = 0; i < 6; ++i) si.pad[i] = 0;
    sh.xp.part.seginfo[(2u << (p - 1)) - 2u + part] = si;
}

// Stateless adaptive pass of one partition order over the thread's chunk
// (ref block/encoder.cpp:201-263 with adapt_k_stateless :72-77).  Partial sums leave through `flush`.
// sh.u holds the plain residual (no flag bits) in this phase.
template <class G, bool NARROW, bool ZR = true, class Flush>
LACX_HD void partition_pass(const Thread<G>& th, const Smem<G>& sh, int p, Flush&& flush) {
    if (th.cnt <= 0) return;
    const uint32_t n = th.n;
    const uint32_t base = n >> p, parts = 1u << p;
    uint32_t part = (uint32_t)th.a / base;
    if (part >= parts) part = parts - 1u;
    uint32_t s = part * base;
    uint32_t e = (part + 1u == parts) ? n : s + base;
    const uint32_t segbase = (2u << (p - 1)) - 2u;
    uint64_t P = sh.tabP[th.tid];  // P_{a-1}
    uint64_t Pseg;                 // P_{s-1}
    {
        const uint32_t cs = s / G::CH;
        Pseg = sh.tabP[cs];
        for (uint32_t j = cs * G::CH; j < s; ++j) Pseg += sh.u[sw<G>((int)j)];
    }
    int32_t lastnz = sh.tabNZ[th.tid];
    if (lastnz < (int32_t)s - 1) lastnz = (int32_t)s - 1;
    int32_t f = th.a - 1 - lastnz;
    unsigned long long rice = 0, bin = 0, zr = 0;
    uint32_t hasrun = 0;
    uint32_t u = sh.u[th.tid];
    uint32_t x1 = peek_u<G>(sh, (uint32_t)th.a + 1u, n), x2 = peek_u<G>(sh, (uint32_t)th.a + 2u, n),
             x3 = peek_u<G>(sh, (uint32_t)th.a + 3u, n);
    for (int i = 0; i < th.cnt; ++i) {
        const uint32_t j = (uint32_t)(th.a + i);
        if (j == e) {  // partition boundary inside the chunk
            flush(segbase + part, rice, bin, zr, hasrun);
            rice = bin = zr = 0;
            hasrun = 0;
            ++part;
            s = e;
            e = (part + 1u == parts) ? n : s + base;
            Pseg = P;
            f = 0;
        }
        const uint32_t kin = (j == s) ? (uint32_t)sh.xp.part.seginfo[segbase + part].ak
                                      : kmean_t<NARROW>(P - Pseg, j - s);
        const uint32_t rc = ((kin >= 31u) ? 0u : (u >> kin)) + 1u + kin;
        rice += rc;
        bin += (u == 0) ? 2u : ((u <= 4u) ? 3u : 2u + rc);
        if (ZR) {
            const bool z = (u == 0);
            f = z ? f + 1 : 0;
            const uint32_t n1 = (j + 1u < e) ? x1 : 1u;
            const uint32_t n2 = (j + 2u < e) ? x2 : 1u;
            const uint32_t n3 = (j + 3u < e) ? x3 : 1u;
            const int ahead = (n1 != 0) ? 0 : ((n2 != 0) ? 1 : ((n3 != 0) ? 2 : 3));
            const bool in4 = z && (f + ahead >= 4);
            if (!in4) {
                const uint32_t esc = 1u << ((kin + 3u) < 24u ? (kin + 3u) : 24u);
                zr += 2u + ((u > esc) ? 32u : rc);
            } else if (n1 != 0) {
                zr += 2u + (((uint32_t)(f - 4)) >> 2) + 3u;
                hasrun = 1;
            }
        }
        P += u;
        u = x1;
        x1 = x2;
        x2 = x3;
        x3 = peek_u<G>(sh, (uint32_t)(th.a + i) + 4u, n);
    }
    flush(segbase + part, rice, bin, zr, hasrun);
}


// True when every partition boundary of every order 1..max_p falls on a chunk boundary, i.e. each
// thread's chunk lies inside exactly one partition of each order (all 16384-sample blocks and the
// 256-sample probes).  The fused pass below then evaluates all orders in one walk over the chunk.
template <class G>
LACX_HD bool partitions_chunk_aligned(uint32_t n, int max_p) {
    return max_p > 0 && (n % ((uint32_t)G::CH << max_p)) == 0u;
}

// All partition orders in one pass (32-bit arithmetic: requires total sum of u < kNarrowLimit).
// Same numbers as partition_pass<G, true> run for p = 1..max_p.
template <class G, bool ZR = true, class Flush>
LACX_HD void partition_fused(const Thread<G>& th, const Smem<G>& sh, int max_p, Flush&& flush) {
    // Every lane of the wave reaches the flush (its sums go through wave-wide reductions on the device): a chunk beyond
    // the slot contributes zeros.
    const bool live = th.cnt > 0;
    const uint32_t n = th.n;
    const int t = th.tid;
    const uint32_t a = (uint32_t)th.a;
    const uint32_t Pa = (uint32_t)sh.tabP[t];  // P_{a-1}
    uint32_t s[G::MAXP], rem[G::MAXP], Pseg[G::MAXP], ak[G::MAXP], sidx[G::MAXP];
    uint32_t rice[G::MAXP], bin[G::MAXP], zr[G::MAXP];
    uint32_t hasrun = 0;  // bit p-1
#pragma unroll
    for (int q = 0; q < G::MAXP; ++q) {
        const int p = q + 1;
        rice[q] = bin[q] = zr[q] = 0;
        s[q] = rem[q] = Pseg[q] = ak[q] = sidx[q] = 0;
        if (p <= max_p) {
            const uint32_t base = n >> p;
            const uint32_t part = !live ? 0u : ((n & (n - 1u)) == 0u ? a >> (31 - clz32(n) - p) : a / base);
            s[q] = part * base;
            rem[q] = s[q] + base - a;  // samples from a to the end of the partition
            Pseg[q] = (uint32_t)sh.tabP[s[q] / (uint32_t)G::CH];
            sidx[q] = (2u << (p - 1)) - 2u + part;
            ak[q] = sh.xp.part.seginfo[sidx[q]].ak;
        }
    }
    // Orders whose partition containing this chunk starts at the same sample as the next lower order's see the same
    // prefix, the same sample counts and the same initial k, hence the same Rice and bin costs sample for sample (the
    // zero-run cost also depends on where the partition ends, so this holds for the ZR = false instance only).
    // Partitions of the low orders span whole waves: when every lane agrees, the order is skipped and takes its
    // sums from the lower one afterwards (about 1.5 of 8 orders on a 16384-sample block).
    uint32_t dup = 0;  // bit q: order q + 1 repeats order q (wave-uniform)
    if (!ZR) {
#pragma unroll
        for (int q = 1; q < G::MAXP; ++q)
            if (q < max_p && wave_all(s[q] == s[q - 1] && ak[q] == ak[q - 1])) dup |= 1u << q;
    }
    int32_t fg = (int32_t)a - 1 - sh.tabNZ[t];  // zeros ending just before the chunk (not yet clipped to a partition)
    uint32_t P = Pa;
    uint32_t u = sh.u[t];
    uint32_t x1 = peek_u<G>(sh, a + 1u, n), x2 = peek_u<G>(sh, a + 2u, n), x3 = peek_u<G>(sh, a + 3u, n);
    // One sample of the chunk for every order.  FIRST: the chunk's first sample, the only one that can be the first
    // of a partition (partitions start on chunk boundaries here) and then takes the partition's initial k.
    // kin <= 30 throughout (see kmean32), so u >> kin needs no k >= 31 special case.
    auto sample = [&](int i, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const uint32_t j = a + (uint32_t)i;
        const bool z = (u == 0);
        fg = z ? fg + 1 : 0;
        const uint32_t ahead_g = (x1 != 0) ? 0u : ((x2 != 0) ? 1u : ((x3 != 0) ? 2u : 3u));
        const uint32_t small = (u == 0) ? 2u : 3u;
        const bool is_small = u <= 4u;
#pragma unroll
        for (int q = 0; q < G::MAXP; ++q) {
            if (q < max_p && !((dup >> q) & 1u)) {
                const uint32_t cbefore = j - s[q];                       // samples of the partition before j
                uint32_t kin = kmean32(P - Pseg[q], FIRST ? (cbefore ? cbefore : 1u) : cbefore);
                if (FIRST) kin = (cbefore == 0) ? ak[q] : kin;
                const uint32_t rc = (u >> kin) + 1u + kin;
                rice[q] += rc;
                bin[q] += is_small ? small : 2u + rc;
                if (ZR) {
                    const uint32_t left = rem[q] - 1u - (uint32_t)i;          // samples after j inside the partition
                    const uint32_t fp = ((uint32_t)fg < cbefore + 1u) ? (uint32_t)fg : cbefore + 1u;
                    const uint32_t ahead = ahead_g < left ? ahead_g : left;
                    const bool in4 = z & (fp + ahead >= 4u);
                    const bool runend = in4 & (ahead == 0u);
                    const uint32_t esc = 1u << ((kin + 3u) < 24u ? (kin + 3u) : 24u);
                    const uint32_t plain = 2u + ((u > esc) ? 32u : rc);
                    const uint32_t token = 5u + ((fp - 4u) >> 2);
                    zr[q] += in4 ? (runend ? token : 0u) : plain;
                    hasrun |= runend ? (1u << q) : 0u;
                }
            }
        }
        P += u;
        u = x1;
        x1 = x2;
        x2 = x3;
        x3 = peek_u<G>(sh, j + 4u, n);
    };
    if (live) {
        sample(0, std::true_type{});
        for (int i = 1; i < th.cnt; ++i) sample(i, std::false_type{});
    }
#pragma unroll
    for (int q = 1; q < G::MAXP; ++q) {
        if ((dup >> q) & 1u) {  // ascending: a run of repeated orders cascades
            rice[q] = rice[q - 1];
            bin[q] = bin[q - 1];
        }
    }
#pragma unroll
    for (int q = 0; q < G::MAXP; ++q) {
        if (q < max_p) flush(q, sidx[q], rice[q], bin[q], zr[q], (hasrun >> q) & 1u);
    }
}

// ---------------------------------------------------------------------------------------------
// Partition search without walking the samples (no zero-run costs; 32-bit sums; partitions on chunk boundaries).
// Inside a partition the Rice parameter of sample j is kmean(S_j, c_j) of the prefix sum S_j and the count c_j of the
// partition's samples before j (ref block/encoder.cpp:72-77, 201-263).  kmean is monotone: it grows with S and falls
// with c (the smallest k with S < c 2^k + ceil(c / 2)).  Over the samples a+1 .. a+CH-1 of a chunk S only grows and c
// only grows, so every parameter in force there lies between kmean(S_first, c_last) and kmean(S_last, c_first).  Where
// the two agree -- K -- the chunk's costs need no walk: sum (u >> K) comes from the thread's bit-sliced plane counts
// (sum over the slices l of (cs[l] >> K) << l), and the bin code's exceptions (u <= 4) from a census of the small
// values.  A chunk that opens its partition has its first sample costed by itself (it takes the partition's initial
// k).  Chunks where the bounds disagree (about a quarter: the first few of every partition, where the prefix mean
// still moves) are queued per wave and walked by partition_slow_entry, densely packed over the wave's lanes.
// Same numbers as partition_fused<G, false>.
// ---------------------------------------------------------------------------------------------
// Census of the chunk's small values: 5-bit counts of u == 0, 1, 2, 3, 4 in fields 0..4, of u >= 5 in field 5.
LACX_HD uint32_t census_code(uint32_t u) { return 1u << (5u * (u < 5u ? u : 5u)); }

template <class G>
LACX_HD uint32_t small_census(const Thread<G>& th, const Smem<G>& sh, uint32_t* u_first, uint32_t* zmask) {
    uint32_t cen = 0, zm = 0;
#pragma unroll
    for (int i = 0; i < G::CH; ++i) {
        const uint32_t u = sh.u[i * G::T + th.tid];
        if (i == 0) *u_first = u;
        cen += census_code(u);
        zm |= flag01(u == 0u) << i;
    }
    *zmask = zm;
    return cen;
}

// sum over the census of (v >> K) for the small values v = 1..4
LACX_HD uint32_t census_shift_sum(uint32_t cen, uint32_t K) {
    const uint32_t n1 = (cen >> 5) & 31u, n2 = (cen >> 10) & 31u, n3 = (cen >> 15) & 31u, n4 = (cen >> 20) & 31u;
    const uint32_t s0 = n1 + 2u * n2 + 3u * n3 + 4u * n4, s1 = n2 + n3 + 2u * n4, s2 = n4;
    return K == 0u ? s0 : (K == 1u ? s1 : (K == 2u ? s2 : 0u));
}

template <class G, class Flush, class Enqueue>
LACX_HD void partition_quick(const Thread<G>& th, const Smem<G>& sh, int max_p, Flush&& flush, Enqueue&& enqueue) {
    // (chunks are complete or empty here; an empty chunk still takes part in the wave-wide steps of `enqueue`)
    const bool live = th.cnt > 0;
    const uint32_t n = th.n;
    const int t = th.tid;
    const uint32_t a = (uint32_t)th.a;
    const uint32_t Pa = (uint32_t)sh.tabP[t];                 // P_{a-1}
    const uint32_t csum = (uint32_t)sh.tabP[t + 1] - Pa;      // sum of the chunk
    uint32_t u_first;
    uint32_t zmask_unused;
    const uint32_t cen = small_census(th, sh, &u_first, &zmask_unused);
    const uint32_t u_last = sh.u[(G::CH - 1) * G::T + t];
    const uint32_t code_first = census_code(u_first);
    // a / (n >> p) without a division when n is a power of two (every full block, every probe; block-uniform)
    const bool n_pow2 = (n & (n - 1u)) == 0u;
    const int log2n = 31 - clz32(n);
    for (int q = 0; q < max_p; ++q) {
        const int p = q + 1;
        const uint32_t base = n >> p;
        const uint32_t part = !live ? 0u : (n_pow2 ? a >> (log2n - p) : a / base);
        const uint32_t s = part * base;
        const uint32_t sidx = (2u << (p - 1)) - 2u + part;
        const uint32_t Sa = Pa - (uint32_t)sh.tabP[s / (uint32_t)G::CH];  // sum of the partition's samples before a
        const uint32_t ca = a - s;                                         // ... and their number
        // A chunk that opens its partition (head) has its first sample costed by itself, with the partition's initial
        // k; the constant-parameter test then covers samples a+1 .. a+CH-1, otherwise the whole chunk.
        const bool head = ca == 0u;
        const uint32_t klo = kmean32(head ? u_first : Sa, ca + (uint32_t)G::CH - 1u);
        const uint32_t khi = kmean32(Sa + csum - u_last, head ? 1u : ca);
        const bool ambiguous = live && klo != khi;
        enqueue((uint32_t)t | ((uint32_t)q << 12), ambiguous);
        const bool quick = live && !ambiguous;  // (the others pass zeros: every lane of the wave reaches the flush)
        const uint32_t K = klo;
        uint32_t shifted = 0;  // sum of u >> K over the chunk
#pragma unroll
        for (int l = 0; l < G::LV; ++l) shifted += (th.cs[l] >> K) << l;
        const uint32_t k_a = (uint32_t)sh.xp.part.seginfo[sidx].ak;
        const uint32_t rc_a = (u_first >> k_a) + 1u + k_a;               // head only
        const uint32_t ncon = (uint32_t)G::CH - (head ? 1u : 0u);         // samples costed at K
        const uint32_t cen_con = cen - (head ? code_first : 0u);
        if (head) shifted -= u_first >> K;
        const uint32_t rice = shifted + ncon * (1u + K) + (head ? rc_a : 0u);
        // bin: 2 for a zero, 3 for 1..4, else 2 + the Rice code
        const uint32_t n14 = ((cen_con >> 5) & 31u) + ((cen_con >> 10) & 31u) + ((cen_con >> 15) & 31u) + ((cen_con >> 20) & 31u);
        const uint32_t nbig = (cen_con >> 25) & 31u;
        const uint32_t bin_a = 2u + (u_first <= 4u ? (u_first < 1u ? u_first : 1u) : rc_a);
        const uint32_t bin = 2u * ncon + n14 + (shifted - census_shift_sum(cen_con, K)) + nbig * (1u + K) + (head ? bin_a : 0u);
        flush(q, sidx, quick ? rice : 0u, quick ? bin : 0u, 0u, 0u);
    }
}

// ---------------------------------------------------------------------------------------------
// The same search for a block WITH a run of >= 4 zeros: the zero-run cost of every partition is needed too.  With the
// parameter constant it is the k-sum plus 3 + K for every sample outside a run of >= 4 zeros plus the tokens of the runs
// that end in the chunk (run_shape: from the zero mask alone) -- unless the partition cuts a run (ref
// block/encoder.cpp:224-247 counts runs per partition).  A run reaching back past the partition's first sample changes,
// past the head chunk, one thing only: the token, when the run ends inside the chunk, counts the run from the
// partition's first sample.  In the head chunk the first samples may drop out of the run, a run going on past the
// partition's last sample ends there, and a sample above the zero-run escape costs 32 bits flat: those (chunk, order)
// pairs are walked.  Classification first (partition_quick_prepare), so that a wave whose pairs mostly need the walk can
// take partition_fused instead; then the costs (partition_quick_costs).
// ---------------------------------------------------------------------------------------------
template <class G>
struct QuickPrep {
    uint32_t Pa, u_first, cen, code_first;
    uint32_t E;
    int32_t f0;
    RunShape rs;
    uint32_t amb;                       // bit q: (chunk, order q + 1) needs the walk
    uint32_t Kpack[(G::MAXP + 3) / 4];  // the constant parameter of order q + 1, a byte each
};

// A wave whose chunks have more ambiguous (chunk, order) pairs than this walks all orders at once instead
// (partition_fused): the walks of that many pairs, packed 64 to a trip, cost more than the one walk of everything.
constexpr uint32_t kQuickMaxPairs = 256;

template <class G>
LACX_HD void partition_quick_prepare(const Thread<G>& th, const Smem<G>& sh, int max_p, QuickPrep<G>& qp) {
    const bool live = th.cnt > 0;
    const uint32_t n = th.n;
    const int t = th.tid;
    const uint32_t a = (uint32_t)th.a;
    qp.Pa = (uint32_t)sh.tabP[t];                              // P_{a-1}
    const uint32_t csum = (uint32_t)sh.tabP[t + 1] - qp.Pa;    // sum of the chunk
    uint32_t zmask;
    qp.cen = small_census(th, sh, &qp.u_first, &zmask);
    const uint32_t u_last = sh.u[(G::CH - 1) * G::T + t];
    qp.code_first = census_code(qp.u_first);
    // zero-run structure of the chunk as it stands in the block (no partition cutting a run)
    qp.E = zmask | (flag01(peek_u<G>(sh, a + (uint32_t)G::CH, n) == 0u) << G::CH) |
           (flag01(peek_u<G>(sh, a + (uint32_t)G::CH + 1u, n) == 0u) << (G::CH + 1)) |
           (flag01(peek_u<G>(sh, a + (uint32_t)G::CH + 2u, n) == 0u) << (G::CH + 2));
    qp.f0 = (int32_t)a - 1 - sh.tabNZ[t];
    qp.rs = run_shape<G::CH>(qp.E, qp.f0);
    uint32_t any = 0;
#pragma unroll
    for (int l = 0; l < G::LV; ++l) any |= th.cs[l];
    const bool n_pow2 = (n & (n - 1u)) == 0u;
    const int log2n = 31 - clz32(n);
    qp.amb = 0;
#pragma unroll
    for (int w = 0; w < (G::MAXP + 3) / 4; ++w) qp.Kpack[w] = 0;
#pragma unroll
    for (int q = 0; q < G::MAXP; ++q) {
        if (q >= max_p) continue;
        const int p = q + 1;
        const uint32_t base = n >> p;
        const uint32_t part = !live ? 0u : (n_pow2 ? a >> (log2n - p) : a / base);
        const uint32_t s = part * base;
        const uint32_t Sa = qp.Pa - (uint32_t)sh.tabP[s / (uint32_t)G::CH];  // sum of the partition's samples before a
        const uint32_t ca = a - s;                                            // ... and their number
        const bool head = ca == 0u;
        const uint32_t klo = kmean32(head ? qp.u_first : Sa, ca + (uint32_t)G::CH - 1u);
        const uint32_t khi = kmean32(Sa + csum - u_last, head ? 1u : ca);
        const bool cut_front = (qp.E & 1u) != 0u && (uint32_t)qp.f0 > ca;  // opens with a zero of a run that began before the partition
        const bool cut_back = ca + (uint32_t)G::CH == base && ((qp.E >> (G::CH - 1)) & 3u) == 3u;  // last chunk, its last sample and the next one zero
        const bool escape = (any >> ((klo + 3u) < 24u ? (klo + 3u) : 24u)) != 0u;
        const bool ambiguous = klo != khi || (cut_front && head) || cut_back || escape;
        qp.Kpack[q >> 2] |= klo << (8 * (q & 3));
        qp.amb |= (live && ambiguous) ? (1u << q) : 0u;
    }
}

template <class G, class Flush, class Enqueue>
LACX_HD void partition_quick_costs(const Thread<G>& th, const Smem<G>& sh, int max_p, const QuickPrep<G>& qp, Flush&& flush,
                                   Enqueue&& enqueue) {
    // (chunks are complete or empty here; an empty chunk still takes part in the wave-wide steps of `enqueue`)
    const bool live = th.cnt > 0;
    const uint32_t n = th.n;
    const int t = th.tid;
    const uint32_t a = (uint32_t)th.a;
    const bool n_pow2 = (n & (n - 1u)) == 0u;
    const int log2n = 31 - clz32(n);
    const uint32_t open_run = (uint32_t)ctz32(~qp.E);  // zeros the chunk opens with (counting on into the three samples after it)
#pragma unroll
    for (int q = 0; q < G::MAXP; ++q) {
        if (q >= max_p) continue;
        const int p = q + 1;
        const uint32_t base = n >> p;
        const uint32_t part = !live ? 0u : (n_pow2 ? a >> (log2n - p) : a / base);
        const uint32_t sidx = (2u << (p - 1)) - 2u + part;
        const uint32_t ca = a - part * base;
        const bool head = ca == 0u;
        const bool ambiguous = ((qp.amb >> q) & 1u) != 0u;
        enqueue((uint32_t)t | ((uint32_t)q << 12), ambiguous);
        const bool quick = live && !ambiguous;  // (the others pass zeros: every lane of the wave reaches the flush)
        const uint32_t K = (qp.Kpack[q >> 2] >> (8 * (q & 3))) & 0xFFu;
        uint32_t shifted = 0;  // sum of u >> K over the chunk
#pragma unroll
        for (int l = 0; l < G::LV; ++l) shifted += (th.cs[l] >> K) << l;
        const uint32_t k_a = (uint32_t)sh.xp.part.seginfo[sidx].ak;
        const uint32_t rc_a = (qp.u_first >> k_a) + 1u + k_a;             // head only
        const uint32_t ncon = (uint32_t)G::CH - (head ? 1u : 0u);         // samples costed at K
        const uint32_t cen_con = qp.cen - (head ? qp.code_first : 0u);
        if (head) shifted -= qp.u_first >> K;
        const uint32_t rice = shifted + ncon * (1u + K) + (head ? rc_a : 0u);
        // bin: 2 for a zero, 3 for 1..4, else 2 + the Rice code
        const uint32_t n14 = ((cen_con >> 5) & 31u) + ((cen_con >> 10) & 31u) + ((cen_con >> 15) & 31u) + ((cen_con >> 20) & 31u);
        const uint32_t nbig = (cen_con >> 25) & 31u;
        const uint32_t bin_a = 2u + (qp.u_first <= 4u ? (qp.u_first < 1u ? qp.u_first : 1u) : rc_a);
        const uint32_t bin = 2u * ncon + n14 + (shifted - census_shift_sum(cen_con, K)) + nbig * (1u + K) + (head ? bin_a : 0u);
        // zero-run: the head, when it is not inside a run, pays 2 + its own code at the initial k (32 bits flat above that
        // k's escape); a token of a cut run counts the run from the partition's first sample (see above)
        const bool head_plain = head && qp.rs.first_in4 == 0u;
        const uint32_t esc_a = 1u << ((k_a + 3u) < 24u ? (k_a + 3u) : 24u);
        const uint32_t plain_a = 2u + (qp.u_first > esc_a ? 32u : rc_a);
        const uint32_t nplain = (uint32_t)G::CH - qp.rs.nin4 - (head_plain ? 1u : 0u);  // samples outside runs, costed at K
        uint32_t tokens = qp.rs.tokens;
        if ((qp.E & 1u) != 0u && (uint32_t)qp.f0 > ca && !head && open_run <= (uint32_t)G::CH)
            tokens = tokens + ((ca + open_run - 4u) >> 2) - (((uint32_t)qp.f0 + open_run - 4u) >> 2);
        const uint32_t zr = shifted + nplain * (3u + K) + (head_plain ? plain_a : 0u) + tokens;
        flush(q, sidx, quick ? rice : 0u, quick ? bin : 0u, quick ? zr : 0u, quick ? qp.rs.hasrun : 0u);
    }
}

// One queued (chunk, order) pair of partition_quick: the plain walk over the chunk's samples (32-bit sums, the whole
// chunk inside one partition; ZR: with the zero-run cost, runs cut at the partition's ends).
template <class G, bool ZR = false, class Flush>
LACX_HD void partition_slow_entry(const Smem<G>& sh, uint32_t n, uint32_t entry, Flush&& flush) {
    const uint32_t t = entry & 0xFFFu;
    const int p = (int)(entry >> 12) + 1;
    const uint32_t a = t * (uint32_t)G::CH;
    const uint32_t base = n >> p;
    const uint32_t part = (n & (n - 1u)) == 0u ? a >> (31 - clz32(n) - p) : a / base;  // (block-uniform choice)
    const uint32_t s = part * base;
    const uint32_t sidx = (2u << (p - 1)) - 2u + part;
    uint32_t S = (uint32_t)sh.tabP[t] - (uint32_t)sh.tabP[s / (uint32_t)G::CH];  // sum of the partition's samples before j
    uint32_t c = a - s;                                                           // ... and their number
    uint32_t k = c == 0u ? (uint32_t)sh.xp.part.seginfo[sidx].ak : kmean32(S, c ? c : 1u);
    uint32_t rice = 0, bin = 0, zr = 0, hasrun = 0;
    uint32_t u = sh.u[t];
    int32_t fg = (int32_t)a - 1 - sh.tabNZ[t];  // zeros ending just before the chunk (not yet clipped to the partition)
    uint32_t x1 = 1, x2 = 1, x3 = 1;
    if (ZR) {
        x1 = peek_u<G>(sh, a + 1u, n);
        x2 = peek_u<G>(sh, a + 2u, n);
        x3 = peek_u<G>(sh, a + 3u, n);
    }
#pragma unroll
    for (int i = 0; i < G::CH; ++i) {
        const uint32_t unext = ZR ? x1 : sh.u[((i + 1) & (G::CH - 1)) * G::T + t];
        const uint32_t rc = (u >> k) + 1u + k;
        rice += rc;
        bin += 2u + (u <= 4u ? (u < 1u ? u : 1u) : rc);
        if (ZR) {
            const uint32_t z = flag01(u == 0u);
            fg = (fg + 1) & (int32_t)(0u - z);
            const uint32_t ahead_g = (x1 != 0) ? 0u : ((x2 != 0) ? 1u : ((x3 != 0) ? 2u : 3u));
            const uint32_t left = base - 1u - c;                              // samples after this one inside the partition
            const uint32_t fp = ((uint32_t)fg < c + 1u) ? (uint32_t)fg : c + 1u;
            const uint32_t ahead = ahead_g < left ? ahead_g : left;
            const uint32_t in4 = z & flag01(fp + ahead >= 4u);
            const uint32_t runend = in4 & flag01(ahead == 0u);
            const uint32_t esc = 1u << ((k + 3u) < 24u ? (k + 3u) : 24u);
            const uint32_t plain = 2u + ((u > esc) ? 32u : rc);
            const uint32_t token = 5u + ((fp - 4u) >> 2);
            zr += (plain & (in4 - 1u)) | (token & (0u - runend));
            hasrun |= runend;
            x1 = x2;
            x2 = x3;
            x3 = peek_u<G>(sh, a + (uint32_t)i + 4u, n);
        }
        S += u;
        ++c;
        k = kmean32(S, c);
        u = unext;
    }
    flush(sidx, rice, bin, zr, hasrun);
}

// Mode choice of one partition (ref block/encoder.cpp:495-525); returns bits, writes (mode<<5)|k.
template <class G>
LACX_HD uint64_t seg_choose(Smem<G>& sh, uint32_t idx, int zero_run) {
    PartMem<G>& pm = sh.xp.part;
    const SegInfo si = pm.seginfo[idx];
    const uint64_t normal = pm.segacc[idx][0], bin = pm.segacc[idx][1];
    const bool allow_zr = zero_run && pm.segrun[idx] != 0;
    const uint64_t zr = allow_zr ? pm.segacc[idx][2] : normal;
    uint32_t mode = 0, k = si.ak;
    uint64_t bits = normal;
    if (allow_zr && zr < bits) {
        mode = 1;
        bits = zr;
    }
    if (bin < bits) {
        mode = 2;
        bits = bin;
    }
    if (si.sbits < bits || si.sbits <= bits + bits / 20u) {
        mode = 3;
        k = si.sk;
        bits = si.sbits;
    }
    pm.choice[idx] = (uint8_t)((mode << 5) | k);
    return bits;
}

// Final decisions by thread 0 (ref block/encoder.cpp:421-484, 527-552, 773-795).
template <class G>
LACX_HD void finalize_plan(Smem<G>& sh, uint32_t n, int zero_run, int max_p, ChannelPlan* out) {
    const int cand = sh.best_cand;
    uint8_t ptype, order;
    if (cand <= 4) {
        ptype = 0;
        order = (uint8_t)cand;
    } else if (cand == 5) {
        ptype = 1;
        order = 2;
    } else {
        ptype = 2;
        order = sh.lpc.used[cand - 6];  // chosen_order: used <= cand <= max_valid_order
    }
    out->valid = 1;
    out->predictor_type = ptype;
    out->order = order;
    for (int i = 0; i < 12; ++i) out->coef[i] = (cand >= 6) ? sh.lpc.coef[cand - 6][i + 1] : (int16_t)0;
    // unpartitioned mode (ref :432-456)
    const bool allow_zr = zero_run && sh.best_hasrun;
    uint32_t mode = 0, k = sh.best_k0;
    uint64_t bits = sh.best_rice;
    if (allow_zr && sh.best_zr <= bits) {
        bits = sh.best_zr;
        mode = 1;
    }
    if (sh.best_bin < bits) {
        bits = sh.best_bin;
        mode = 2;
    }
    if (sh.best_static < bits) {
        bits = sh.best_static;
        mode = 3;
        k = sh.best_sk;
    }
    uint64_t best_total = bits + 8u + 7u;
    best_total += (8u - (best_total & 7u)) & 7u;
    int best_p = 0;
    for (int p = 1; p <= max_p; ++p) {
        uint64_t total = sh.xp.part.pbits[p] + 8u + 7ull * (1u << p);
        total += (8u - (total & 7u)) & 7u;
        const uint64_t margin = best_total / 20u;
        if (total < best_total || (total <= best_total + margin && best_p == 0)) {
            best_total = total;
            best_p = p;
        }
    }
    out->partition_order = (uint8_t)best_p;
    out->total_bits = best_total;
    out->payload_bytes = (uint32_t)((16u + (ptype == 2 ? 16u * order : 0u) + best_total) >> 3);
    uint32_t any_zr = 0;  // some partition is coded in zero-run mode (the emit then needs the run-length table)
    if (best_p == 0) {
        out->part_mode_k[0] = (uint8_t)((mode << 5) | k);
        any_zr = mode == 1u;
    } else {
        const uint32_t parts = 1u << best_p, segbase = (2u << (best_p - 1)) - 2u;
        for (uint32_t i = 0; i < parts; ++i) {
            const uint8_t c = sh.xp.part.choice[segbase + i];
            out->part_mode_k[i] = c;
            any_zr |= (uint32_t)((c >> 5) == 1u);
        }
    }
    sh.plan_any_zr = any_zr;
}

}  // namespace lacx
