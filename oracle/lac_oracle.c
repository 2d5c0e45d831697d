/* oracle/lac_oracle.c -- TEST INFRASTRUCTURE ONLY (parity checker + CPU baseline), never the product.
 *
 * Plain-C restatement of the LAC block-encode path of audexdev/Lossless-Audio-Codec (reference tree
 * read-only at /root/reference; every function cites the file:line it follows).  The algorithm is
 * restated loop for loop in the reference's own scalar order -- deliberately NOT the data-parallel
 * formulation the HIP kernels use -- so that a disagreement between the two points at the kernels.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py byte-compares this file's output with the
 * unmodified reference compiled in place (oracle/_ref/liblac_ref.so, recipe in oracle/Makefile) over
 * seeded inputs, and tests/test_golden.py checks it against tests/golden/ (minted from that build).
 * The reference's own tests hold no golden bytes, so those two are the pins.
 *
 * Arithmetic notes: Levinson-Durbin runs in `long double`, which on the x86-64 hosts used here and
 * on the GPU box is the x87 80-bit format the reference's bytes depend on.
 */
#include "lac_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * constants: src/codec/block/constants.hpp:6-15, src/codec/block/encoder.cpp:41-59
 * ---------------------------------------------------------------------------------------------- */
enum {
    kMaxBlock = 16384,
    kZeroRunMin = 4,
    kZeroRunK = 2,
    kMinPartition = 32,
    kMaxPartitionOrder = 8,
    kInitialScan = 256,
    kInitialMaxK = 12,
    kModeRice = 0,
    kModeZr = 1,
    kModeBin = 2,
    kModeStatic = 3,
    kPredFixed = 0,
    kPredFir = 1,
    kPredLpc = 2
};
static const int kOrderCandidates[5] = {4, 6, 8, 10, 12};

void laco_free(void* p) { free(p); }

/* ------------------------------------------------------------------------------------------------
 * MSB-first bit writer: src/codec/bitstream/bit_writer.cpp:15-111
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t* buf;
    size_t len, cap;
    uint8_t cur;
    int pos; /* bits already used in cur, 0..7 */
} bitw;

static void bw_push(bitw* w, uint8_t b) {
    if (w->len == w->cap) {
        w->cap = w->cap ? w->cap * 2 : 256;
        w->buf = (uint8_t*)realloc(w->buf, w->cap);
    }
    w->buf[w->len++] = b;
}
static void bw_bit(bitw* w, uint32_t bit) { /* bit_writer.cpp:15-27 */
    w->cur |= (uint8_t)((bit ? 1u : 0u) << (7 - w->pos));
    if (++w->pos == 8) {
        bw_push(w, w->cur);
        w->cur = 0;
        w->pos = 0;
    }
}
static void bw_bits(bitw* w, uint32_t value, int nbits) { /* bit_writer.cpp:29-70, nbits <= 32 */
    for (int i = nbits - 1; i >= 0; --i) bw_bit(w, (value >> i) & 1u);
}
static void bw_unary(bitw* w, uint32_t ones) { /* bit_writer.cpp:72-88 */
    while (w->pos != 0 && ones > 0) {
        bw_bit(w, 1);
        --ones;
    }
    while (ones >= 8) {
        bw_push(w, 0xFF);
        ones -= 8;
    }
    while (ones > 0) {
        bw_bit(w, 1);
        --ones;
    }
}
static void bw_flush(bitw* w) { /* bit_writer.cpp:105-111 */
    if (w->pos) {
        bw_push(w, w->cur);
        w->cur = 0;
        w->pos = 0;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Rice primitives: src/codec/rice/rice.cpp:7-32, src/codec/block/encoder.cpp:61-87
 * ---------------------------------------------------------------------------------------------- */
static inline uint32_t zigzag(int32_t r) { /* encoder.cpp:61-65 */
    return ((uint32_t)r << 1) ^ (r < 0 ? 0xFFFFFFFFu : 0u);
}
static inline uint64_t rice_bits(uint32_t u, uint32_t k) { /* encoder.cpp:67-70 */
    const uint32_t q = (k >= 31u) ? 0u : (u >> k);
    return (uint64_t)q + 1u + k;
}
static inline uint32_t bit_width64(uint64_t v) {
    uint32_t w = 0;
    while (v) {
        ++w;
        v >>= 1;
    }
    return w;
}
static inline uint32_t adapt_stateless(uint64_t sum, uint32_t count) { /* encoder.cpp:72-77 */
    if (count == 0) return 0;
    const uint64_t mean = (sum + (count >> 1)) / count;
    if (mean <= 1) return 0;
    const uint32_t w = bit_width64(mean - 1u);
    return w < 31u ? w : 31u;
}
static void write_rice_unsigned(bitw* w, uint32_t value, uint32_t k) { /* encoder.cpp:79-87 */
    const uint32_t q = (k >= 31u) ? 0u : (value >> k);
    bw_unary(w, q);
    bw_bit(w, 0);
    if (k > 0) bw_bits(w, value & ((1u << k) - 1u), (int)k);
}
static void rice_encode(bitw* w, int32_t v, uint32_t k) { /* rice.cpp:17-32 */
    const uint32_t u = zigzag(v);
    const uint32_t q = (k >= 32u) ? 0u : (u >> k);
    const uint32_t r = (k >= 32u) ? u : (u & (((uint32_t)1 << k) - 1u));
    bw_unary(w, q);
    bw_bit(w, 0);
    if (k > 0) bw_bits(w, r, (int)k);
}

/* Rice::AdaptState / Rice::adapt_k: src/codec/rice/rice.hpp:15-32, 45-114 */
typedef struct {
    uint64_t previous_sum;
    uint32_t window_index, micro_index, window_filled;
    uint64_t window_sum;
    uint16_t large_q, zero_q;
    uint32_t recent_u[256];
    uint8_t large_flags[96], zero_flags[96];
} adapt_state;

static void adapt_init(adapt_state* s) { memset(s, 0, sizeof(*s)); }

static uint32_t adapt_k(uint64_t sum, uint32_t count, adapt_state* s) {
    if (count == 0) return 0;
    const uint64_t cur = sum - s->previous_sum; /* rice.hpp:49-50 */
    s->previous_sum = sum;
    const uint32_t mi = s->micro_index; /* rice.hpp:53-55 */
    s->large_q = (uint16_t)(s->large_q - s->large_flags[mi]);
    s->zero_q = (uint16_t)(s->zero_q - s->zero_flags[mi]);
    if (s->window_filled < 256u) { /* rice.hpp:58-64 */
        ++s->window_filled;
    } else {
        s->window_sum -= s->recent_u[s->window_index];
    }
    s->recent_u[s->window_index] = (uint32_t)cur;
    s->window_sum += cur;
    const uint64_t mean = (sum + (count >> 1)) / count; /* rice.hpp:68-71 */
    uint32_t k = 0;
    if (mean > 1) {
        k = bit_width64(mean - 1u);
        if (k > 31u) k = 31u;
    }
    const uint32_t q_base = (k >= 31u) ? 0u : (uint32_t)(cur >> k); /* rice.hpp:73-80 */
    const uint8_t is_large = q_base > 3u;
    const uint8_t is_zero = q_base == 0u;
    s->large_q = (uint16_t)(s->large_q + is_large);
    s->zero_q = (uint16_t)(s->zero_q + is_zero);
    s->large_flags[mi] = is_large;
    s->zero_flags[mi] = is_zero;
    int32_t bias = 0; /* rice.hpp:83-94 */
    if (s->window_filled > 0 && mean > 0) {
        const uint64_t local_mean = (s->window_filled == 256u)
                                        ? ((s->window_sum + 128u) >> 8)
                                        : ((s->window_sum + (s->window_filled >> 1)) / s->window_filled);
        if (local_mean * 3 > mean * 4) {
            bias = 1;
        } else if (local_mean * 4 + 3 < mean * 3) {
            bias = -1;
        }
    }
    if (s->window_index + 1 >= 96u || s->window_filled >= 96u) { /* rice.hpp:97-105 */
        const uint32_t ws = (s->window_filled >= 96u) ? 96u : s->window_filled;
        if ((uint32_t)s->large_q * 4 >= ws * 3) {
            bias = (bias + 1 < 1) ? bias + 1 : 1;
        } else if ((uint32_t)s->zero_q * 5 >= ws * 4) {
            bias = (bias - 1 > -1) ? bias - 1 : -1;
        }
    }
    int32_t bk = (int32_t)k + bias; /* rice.hpp:107-113 */
    if (bk < 0) bk = 0;
    if (bk > 31) bk = 31;
    s->micro_index = (s->micro_index + 1u == 96u) ? 0u : s->micro_index + 1u;
    s->window_index = (s->window_index + 1u) & 255u;
    return (uint32_t)bk;
}

void laco_adapt_k_sequence(const uint32_t* u, uint32_t n, uint32_t* k_out) {
    adapt_state st;
    adapt_init(&st);
    uint64_t sum = 0;
    for (uint32_t i = 0; i < n; ++i) {
        sum += u[i];
        k_out[i] = adapt_k(sum, i + 1, &st);
    }
}

/* ------------------------------------------------------------------------------------------------
 * LPC: src/codec/lpc/lpc.cpp
 * ---------------------------------------------------------------------------------------------- */
void laco_autocorr(const int32_t* pcm, uint32_t n, int order, int64_t* r) { /* lpc.cpp:80-96 */
    for (int k = 0; k <= order; ++k) {
        int64_t sum = 0;
        for (uint32_t i = (uint32_t)k; i < n; ++i) sum += (int64_t)pcm[i] * (int64_t)pcm[i - (uint32_t)k];
        r[k] = sum;
    }
}

static int levinson(const long double* R, int order, long double* a) { /* lpc.cpp:98-154 */
    const long double eps = 1e-8L;
    long double E[33], K[33], prevA[33];
    memset(E, 0, sizeof(E));
    memset(K, 0, sizeof(K));
    for (int i = 0; i < 33; ++i) prevA[i] = 0.0L;
    E[0] = R[0];
    if (!isfinite(E[0]) || E[0] < eps) {
        for (int i = 0; i <= order; ++i) a[i] = 0.0L;
        return 0;
    }
    int achieved = 0;
    for (int i = 1; i <= order; ++i) {
        long double acc = 0.0L;
        for (int j = 1; j < i; ++j) acc += prevA[j] * R[i - j];
        const long double denom = E[i - 1];
        if (!isfinite(denom) || denom < eps) break;
        long double ki = (R[i] - acc) / denom;
        if (!isfinite(ki)) break;
        if (ki > 0.999L) ki = 0.999L;
        if (ki < -0.999L) ki = -0.999L;
        K[i] = ki;
        const long double e_new = (1.0L - K[i] * K[i]) * E[i - 1];
        if (!isfinite(e_new) || e_new < eps) {
            achieved = i - 1;
            break;
        }
        a[i] = K[i];
        for (int j = 1; j < i; ++j) a[j] = prevA[j] - K[i] * prevA[i - j];
        for (int j = 1; j <= i; ++j) prevA[j] = a[j];
        E[i] = e_new;
        achieved = i;
    }
    return achieved;
}

static int16_t quantize_q15(double c) { /* lpc.cpp:73-78 */
    double scaled = round(c * 32768.0);
    if (scaled < -32768.0) scaled = -32768.0;
    if (scaled > 32767.0) scaled = 32767.0;
    return (int16_t)scaled;
}

int laco_levinson_q15(const int64_t* r, int order, int16_t* coeffs) { /* lpc.cpp:156-186 */
    long double R[33], a[33];
    for (int i = 0; i <= order; ++i) {
        R[i] = (long double)r[i];
        a[i] = 0.0L;
    }
    if (R[0] < 1.0L) R[0] = 1.0L; /* lpc.cpp:169-172 */
    const int used = levinson(R, order, a);
    coeffs[0] = 0;
    for (int i = 1; i <= used; ++i) coeffs[i] = quantize_q15((double)a[i]);
    for (int i = used + 1; i <= order; ++i) coeffs[i] = 0;
    return used;
}

int laco_lpc_analyze(const int32_t* pcm, uint32_t n, int order, int16_t* coeffs) {
    int64_t r[33];
    if (n == 0) {
        for (int i = 0; i <= order; ++i) r[i] = 0;
    } else {
        laco_autocorr(pcm, n, order, r);
    }
    return laco_levinson_q15(r, order, coeffs);
}

/* lpc.cpp:38-61: open-loop residual; returns 0 if any value leaves int32. */
static int lpc_residual_order(const int32_t* x, uint32_t n, const int16_t* c, int order, int32_t* res) {
    for (uint32_t i = 0; i < n; ++i) {
        int64_t acc = 0;
        const int taps = (order < (int)i) ? order : (int)i;
        for (int t = 1; t <= taps; ++t) acc += (int64_t)c[t] * (int64_t)x[i - (uint32_t)t];
        const int64_t pred = acc >> 15;
        const int64_t diff = (int64_t)x[i] - pred;
        if (diff < INT32_MIN || diff > INT32_MAX) return 0;
        res[i] = (int32_t)diff;
    }
    return 1;
}

/* lpc.cpp:188-229 (with build_residual_attempt_orders :24-36). cand = the LPC object's order. */
static void lpc_compute_residual(const int32_t* x, uint32_t n, const int16_t* c, int cand, int32_t* res,
                                 int* used_inout) {
    int start = *used_inout;
    if (start > cand) start = cand;
    if (start < 0) start = 0;
    int attempts[8];
    int na = 0;
    attempts[na++] = start;
    static const int fb[5] = {12, 10, 8, 6, 4};
    for (int i = 0; i < 5; ++i) {
        if (fb[i] < start && fb[i] <= cand) {
            int dup = 0;
            for (int j = 0; j < na; ++j) dup |= attempts[j] == fb[i];
            if (!dup) attempts[na++] = fb[i];
        }
    }
    {
        int dup = 0;
        for (int j = 0; j < na; ++j) dup |= attempts[j] == 0;
        if (!dup) attempts[na++] = 0;
    }
    for (int ai = 0; ai < na; ++ai) {
        const int o = attempts[ai];
        if (o <= 0) {
            memcpy(res, x, sizeof(int32_t) * n);
            *used_inout = 0;
            return;
        }
        if (lpc_residual_order(x, n, c, o, res)) {
            *used_inout = o;
            return;
        }
    }
    memcpy(res, x, sizeof(int32_t) * n);
    *used_inout = 0;
}

/* ------------------------------------------------------------------------------------------------
 * Block encoder cost model: src/codec/block/encoder.cpp:93-309
 * ---------------------------------------------------------------------------------------------- */
static uint8_t max_partition_order(uint32_t n) { /* encoder.cpp:93-101 */
    uint8_t mp = 0;
    for (uint8_t p = 1; p <= kMaxPartitionOrder; ++p) {
        if ((n >> p) < (uint32_t)kMinPartition) break;
        mp = p;
    }
    return mp;
}

static uint32_t partition_sizes(uint32_t n, uint8_t p, uint32_t* sizes) { /* encoder.cpp:103-119 */
    if (p == 0 || (n >> p) == 0) {
        sizes[0] = n;
        return 1;
    }
    const uint32_t base = n >> p, parts = 1u << p;
    for (uint32_t i = 0; i < parts; ++i) sizes[i] = base;
    sizes[parts - 1] = n - base * (parts - 1u);
    return parts;
}

static uint32_t estimate_initial_k(const int32_t* r, uint32_t n) { /* encoder.cpp:121-158 */
    if (n == 0) return 0;
    const uint32_t count = n < (uint32_t)kInitialScan ? n : (uint32_t)kInitialScan;
    uint64_t cost[kInitialMaxK + 1];
    memset(cost, 0, sizeof(cost));
    uint64_t sum_abs = 0;
    for (uint32_t i = 0; i < count; ++i) {
        const uint32_t u = zigzag(r[i]);
        sum_abs += u;
        for (uint32_t k = 0; k <= (uint32_t)kInitialMaxK; ++k) cost[k] += (uint64_t)(u >> k) + 1u + k;
    }
    uint32_t mean_based = 0;
    {
        const uint64_t mean = (sum_abs + (count >> 1)) / count;
        while (((uint64_t)1u << mean_based) < mean && mean_based < 15u) ++mean_based;
    }
    uint32_t best_k = mean_based;
    uint64_t best = UINT64_MAX;
    for (uint32_t k = 0; k <= (uint32_t)kInitialMaxK; ++k) {
        if (cost[k] < best) {
            best = cost[k];
            best_k = k;
        }
    }
    return best_k < 15u ? best_k : 15u;
}

static uint32_t estimate_static_k(const int32_t* r, uint32_t n) { /* encoder.cpp:160-180 */
    if (n == 0) return 0;
    uint64_t cost[16];
    memset(cost, 0, sizeof(cost));
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t u = zigzag(r[i]);
        for (uint32_t k = 0; k < 16; ++k) cost[k] += rice_bits(u, k);
    }
    uint32_t best_k = 0;
    uint64_t best = UINT64_MAX;
    for (uint32_t k = 0; k < 16; ++k) {
        if (cost[k] < best) {
            best = cost[k];
            best_k = k;
        }
    }
    return best_k;
}

static uint64_t estimate_static_bits(const int32_t* r, uint32_t n, uint32_t k) { /* encoder.cpp:182-188 */
    uint64_t bits = 0;
    for (uint32_t i = 0; i < n; ++i) bits += rice_bits(zigzag(r[i]), k);
    return bits;
}

typedef struct {
    uint64_t rice, zr, bin;
    int has_run;
} res_costs;

static res_costs estimate_costs(const int32_t* r, uint32_t n, uint32_t initial_k, int stateless) {
    /* encoder.cpp:201-263 */
    res_costs c = {0, 0, 0, 0};
    if (n == 0) return c;
    uint32_t k = initial_k;
    uint64_t sum = 0;
    uint32_t count = 0;
    adapt_state st;
    if (!stateless) adapt_init(&st);
    uint32_t idx = 0;
    while (idx < n) {
        uint32_t run = 0;
        while (idx + run < n && r[idx + run] == 0) ++run;
        if (run >= (uint32_t)kZeroRunMin) {
            c.has_run = 1;
            c.zr += 2;
            c.zr += rice_bits(run - kZeroRunMin, kZeroRunK);
            for (uint32_t j = 0; j < run; ++j) {
                c.rice += rice_bits(0, k);
                c.bin += 2;
                ++count;
                k = stateless ? adapt_stateless(sum, count) : adapt_k(sum, count, &st);
            }
            idx += run;
            continue;
        }
        const int32_t v = r[idx];
        const uint32_t u = zigzag(v);
        c.rice += rice_bits(u, k);
        if (v == 0) {
            c.bin += 2;
        } else if (v == 1 || v == -1 || v == 2 || v == -2) {
            c.bin += 3;
        } else {
            c.bin += 2 + rice_bits(u, k);
        }
        const uint32_t esc = 1u << ((k + 3u) < 24u ? (k + 3u) : 24u);
        c.zr += 2;
        c.zr += (u > esc) ? 32 : rice_bits(u, k);
        sum += u;
        ++count;
        k = stateless ? adapt_stateless(sum, count) : adapt_k(sum, count, &st);
        ++idx;
    }
    return c;
}

static void fixed_residual(const int32_t* x, uint32_t n, int order, int32_t* res) { /* encoder.cpp:265-295 */
    if (order == 0) {
        memcpy(res, x, sizeof(int32_t) * n);
        return;
    }
    for (uint32_t i = 0; i < (uint32_t)order && i < n; ++i) res[i] = x[i];
    for (uint32_t i = (uint32_t)order; i < n; ++i) {
        int64_t pred = 0;
        switch (order) {
            case 1: pred = x[i - 1]; break;
            case 2: pred = 2LL * x[i - 1] - x[i - 2]; break;
            case 3: pred = 3LL * x[i - 1] - 3LL * x[i - 2] + x[i - 3]; break;
            case 4: pred = 4LL * x[i - 1] - 6LL * x[i - 2] + 4LL * x[i - 3] - x[i - 4]; break;
            default: break;
        }
        res[i] = (int32_t)((int64_t)x[i] - pred);
    }
}

static void fir_residual(const int32_t* x, uint32_t n, int32_t* res) { /* encoder.cpp:297-309, taps {3,-1}>>2 */
    for (uint32_t i = 0; i < 2 && i < n; ++i) res[i] = x[i];
    for (uint32_t i = 2; i < n; ++i) {
        int64_t pred = 3LL * (int64_t)x[i - 1] - (int64_t)x[i - 2];
        pred >>= 2;
        res[i] = (int32_t)((int64_t)x[i] - pred);
    }
}

typedef struct {
    uint8_t type;
    int order_param, used_order;
    uint64_t rice, zr, bin, stat, best;
    uint32_t initial_k, static_k;
    int has_run;
    int16_t coeffs[13];
    int32_t* residual; /* owned scratch */
} pred_eval;

static void score(pred_eval* ev, uint32_t n, int zero_run) { /* encoder.cpp:337-351 */
    ev->initial_k = estimate_initial_k(ev->residual, n);
    const res_costs c = estimate_costs(ev->residual, n, ev->initial_k, 0);
    ev->rice = c.rice;
    ev->has_run = c.has_run;
    ev->zr = (zero_run && c.has_run) ? c.zr : c.rice;
    ev->bin = c.bin;
    ev->static_k = estimate_static_k(ev->residual, n);
    ev->stat = estimate_static_bits(ev->residual, n, ev->static_k);
    uint64_t a = ev->rice < ev->stat ? ev->rice : ev->stat;
    uint64_t b = ev->zr < ev->bin ? ev->zr : ev->bin;
    ev->best = a < b ? a : b;
}

typedef struct {
    uint8_t mode;
    uint32_t k;
    uint64_t bits;
    uint32_t length;
} part_choice;

/* The analysis half of Block::Encoder::encode (encoder.cpp:313-552). Fills `plan`; returns the
 * winning residual in `best_res` (caller provides n int32). */
static void analyze_block(const int32_t* x, uint32_t n, int zero_run, int partitioning, laco_plan* plan,
                          int32_t* best_res) {
    const int max_valid_order = (n > 1) ? (int)((n - 1 < 32u) ? n - 1 : 32u) : 0;
    int32_t* scratch = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
    pred_eval best;
    memset(&best, 0, sizeof(best));
    int have_best = 0;
    pred_eval ev;

#define CONSIDER()                                                                          \
    do {                                                                                    \
        if (!have_best || ev.best < best.best || (ev.best == best.best && ev.type < best.type)) { \
            best = ev;                                                                      \
            memcpy(best_res, scratch, sizeof(int32_t) * n);                                 \
            have_best = 1;                                                                  \
        }                                                                                   \
    } while (0)

    for (int fo = 0; fo <= 4; ++fo) { /* encoder.cpp:362-369 */
        memset(&ev, 0, sizeof(ev));
        ev.type = kPredFixed;
        ev.order_param = fo;
        ev.residual = scratch;
        fixed_residual(x, n, fo, scratch);
        score(&ev, n, zero_run);
        CONSIDER();
    }
    { /* encoder.cpp:372-379 */
        memset(&ev, 0, sizeof(ev));
        ev.type = kPredFir;
        ev.order_param = 2;
        ev.residual = scratch;
        fir_residual(x, n, scratch);
        score(&ev, n, zero_run);
        CONSIDER();
    }
    for (int ci = 0; ci < 5; ++ci) { /* encoder.cpp:382-407 */
        const int cand = kOrderCandidates[ci];
        if (cand > max_valid_order) continue;
        memset(&ev, 0, sizeof(ev));
        ev.type = kPredLpc;
        ev.order_param = cand;
        ev.residual = scratch;
        ev.used_order = laco_lpc_analyze(x, n, cand, ev.coeffs);
        if (ev.used_order == 0) continue;
        lpc_compute_residual(x, n, ev.coeffs, cand, scratch, &ev.used_order);
        if (ev.used_order == 0) continue;
        score(&ev, n, zero_run);
        CONSIDER();
    }
    if (!have_best) { /* encoder.cpp:410-417 (unreachable: fixed-0 is always considered) */
        memset(&ev, 0, sizeof(ev));
        ev.residual = scratch;
        memcpy(scratch, x, sizeof(int32_t) * n);
        score(&ev, n, zero_run);
        CONSIDER();
    }
#undef CONSIDER
    free(scratch);

    int chosen_order = best.order_param; /* encoder.cpp:421-423 */
    if (best.type == kPredLpc) {
        chosen_order = best.used_order < max_valid_order ? best.used_order : max_valid_order;
        if (chosen_order < 1) chosen_order = 1;
    }

    /* unpartitioned choice: encoder.cpp:432-473 */
    const int allow_zr_global = zero_run && best.has_run;
    uint8_t base_mode = kModeRice;
    uint64_t base_bits = best.rice;
    if (allow_zr_global && best.zr <= base_bits) {
        base_bits = best.zr;
        base_mode = kModeZr;
    }
    if (best.bin < base_bits) {
        base_bits = best.bin;
        base_mode = kModeBin;
    }
    uint32_t base_k = best.initial_k;
    if (best.stat < base_bits) {
        base_bits = best.stat;
        base_mode = kModeStatic;
        base_k = best.static_k;
    }

    static const uint32_t zero_sizes = 0;
    (void)zero_sizes;
    part_choice* best_parts = (part_choice*)malloc(sizeof(part_choice) * LACO_MAX_PARTS);
    part_choice* choices = (part_choice*)malloc(sizeof(part_choice) * LACO_MAX_PARTS);
    uint32_t best_count = 1;
    best_parts[0].mode = base_mode;
    best_parts[0].k = base_k;
    best_parts[0].bits = base_bits;
    best_parts[0].length = n;
    uint8_t best_p = 0;
    uint64_t best_total = base_bits + 8 + 7; /* encoder.cpp:475-484 */
    best_total += (8u - (best_total & 7u)) & 7u;

    if (partitioning && n >= (uint32_t)kMinPartition) { /* encoder.cpp:486-552 */
        const uint8_t max_p = max_partition_order(n);
        uint32_t sizes[LACO_MAX_PARTS];
        for (uint8_t p = 1; p <= max_p; ++p) {
            const uint32_t parts = partition_sizes(n, p, sizes);
            uint64_t bits_sum = 0;
            uint32_t offset = 0;
            for (uint32_t pi = 0; pi < parts; ++pi) {
                const uint32_t len = sizes[pi];
                const int32_t* seg = best_res + offset;
                const uint32_t ak = estimate_initial_k(seg, len);
                const uint32_t sk = estimate_static_k(seg, len);
                const res_costs c = estimate_costs(seg, len, ak, 1);
                const uint64_t sbits = estimate_static_bits(seg, len, sk);
                const int allow_zr = zero_run && c.has_run;
                const uint64_t zr_bits = allow_zr ? c.zr : c.rice;
                part_choice pc;
                pc.length = len;
                pc.k = ak;
                pc.mode = kModeRice;
                pc.bits = c.rice;
                if (allow_zr && zr_bits < pc.bits) {
                    pc.mode = kModeZr;
                    pc.bits = zr_bits;
                }
                if (c.bin < pc.bits) {
                    pc.mode = kModeBin;
                    pc.bits = c.bin;
                }
                if (sbits < pc.bits || sbits <= pc.bits + pc.bits / 20u) { /* :518, :190-192 */
                    pc.k = sk;
                    pc.mode = kModeStatic;
                    pc.bits = sbits;
                }
                bits_sum += pc.bits;
                choices[pi] = pc;
                offset += len;
            }
            uint64_t total = bits_sum + 8 + 7ull * parts;
            total += (8u - (total & 7u)) & 7u;
            const uint64_t margin = best_total / 20u;
            if (total < best_total || (total <= best_total + margin && best_p == 0) ||
                (total == best_total && p < best_p)) { /* encoder.cpp:537-544 */
                best_total = total;
                memcpy(best_parts, choices, sizeof(part_choice) * parts);
                best_count = parts;
                best_p = p;
            }
        }
    }

    memset(plan, 0, sizeof(*plan));
    plan->predictor_type = best.type;
    plan->order = (uint8_t)chosen_order;
    plan->partition_order = best_p;
    memcpy(plan->coeffs_q15, best.coeffs, sizeof(best.coeffs));
    plan->part_count = best_count;
    plan->total_bits = best_total;
    plan->best_bits = best.best;
    for (uint32_t i = 0; i < best_count; ++i) {
        plan->part_mode[i] = best_parts[i].mode;
        plan->part_k[i] = (uint8_t)best_parts[i].k;
    }
    free(best_parts);
    free(choices);
}

/* The emit half of Block::Encoder::encode (encoder.cpp:554-838). */
static void emit_block(const laco_plan* plan, const int32_t* res, uint32_t n, bitw* w) {
    const int stateless = plan->partition_order > 0; /* encoder.cpp:556 */
    uint8_t control = (uint8_t)((plan->part_mode[0] & 3u) << 5); /* encoder.cpp:773-778 */
    if (plan->partition_order > 0) control |= 0x80u | (plan->partition_order & 0x0Fu);
    bw_bits(w, plan->predictor_type, 8); /* encoder.cpp:783-795 */
    bw_bits(w, plan->order, 8);
    if (plan->predictor_type == kPredLpc) {
        for (int i = 1; i <= plan->order; ++i) bw_bits(w, (uint16_t)plan->coeffs_q15[i], 16);
    }
    bw_bits(w, control, 8);
    for (uint32_t i = 0; i < plan->part_count; ++i) {
        bw_bits(w, plan->part_mode[i], 2);
        bw_bits(w, plan->part_k[i], 5);
    }
    uint32_t sizes[LACO_MAX_PARTS];
    const uint32_t parts = partition_sizes(n, plan->partition_order, sizes);
    uint32_t offset = 0;
    for (uint32_t pi = 0; pi < parts; ++pi) {
        const uint32_t len = sizes[pi];
        const int32_t* seg = res + offset;
        const uint32_t k0 = plan->part_k[pi];
        const uint8_t mode = plan->part_mode[pi];
        adapt_state st;
        if (!stateless) adapt_init(&st);
        uint32_t k = k0;
        uint64_t sum = 0;
        uint32_t count = 0;
        if (mode == kModeRice) { /* encoder.cpp:585-600 */
            for (uint32_t i = 0; i < len; ++i) {
                rice_encode(w, seg[i], k);
                sum += zigzag(seg[i]);
                k = stateless ? adapt_stateless(sum, i + 1) : adapt_k(sum, i + 1, &st);
            }
        } else if (mode == kModeStatic) { /* encoder.cpp:602-607 */
            for (uint32_t i = 0; i < len; ++i) write_rice_unsigned(w, zigzag(seg[i]), k0);
        } else if (mode == kModeBin) { /* encoder.cpp:609-667 */
            for (uint32_t i = 0; i < len; ++i) {
                const int32_t v = seg[i];
                const uint32_t u = zigzag(v);
                if (v == 0) {
                    bw_bits(w, 0, 2);
                } else if (v == 1 || v == -1) {
                    bw_bits(w, 1, 2);
                    bw_bit(w, v < 0);
                } else if (v == 2 || v == -2) {
                    bw_bits(w, 2, 2);
                    bw_bit(w, v < 0);
                } else {
                    bw_bits(w, 3, 2);
                    rice_encode(w, v, k);
                }
                sum += u;
                ++count;
                k = stateless ? adapt_stateless(sum, count) : adapt_k(sum, count, &st);
            }
        } else { /* zero-run: encoder.cpp:669-771 */
            uint32_t idx = 0;
            while (idx < len) {
                uint32_t run = 0;
                while (idx + run < len && seg[idx + run] == 0) ++run;
                if (run >= (uint32_t)kZeroRunMin) {
                    bw_bits(w, 1, 2);
                    write_rice_unsigned(w, run - kZeroRunMin, kZeroRunK);
                    if (stateless) {
                        count += run;
                        k = adapt_stateless(sum, count);
                    } else {
                        for (uint32_t j = 0; j < run; ++j) {
                            ++count;
                            k = adapt_k(sum, count, &st);
                        }
                    }
                    idx += run;
                    continue;
                }
                const uint32_t u = zigzag(seg[idx]);
                const uint32_t esc = 1u << ((k + 3u) < 24u ? (k + 3u) : 24u);
                if (u > esc) {
                    bw_bits(w, 2, 2);
                    bw_bits(w, u, 32);
                } else {
                    bw_bits(w, 0, 2);
                    rice_encode(w, seg[idx], k);
                }
                sum += u;
                ++count;
                k = stateless ? adapt_stateless(sum, count) : adapt_k(sum, count, &st);
                ++idx;
            }
        }
        offset += len;
    }
    bw_flush(w); /* encoder.cpp:822 */
}

static void block_encode_into(const int32_t* pcm, uint32_t n, int zero_run, int partitioning, bitw* w,
                              laco_plan* plan_out) {
    laco_plan plan;
    int32_t* res = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
    analyze_block(pcm, n, zero_run, partitioning, &plan, res);
    emit_block(&plan, res, n, w);
    if (plan_out) *plan_out = plan;
    free(res);
}

int laco_block_encode(const int32_t* pcm, uint32_t n, int zero_run, int partitioning, uint8_t** out,
                      uint64_t* out_size) {
    bitw w;
    memset(&w, 0, sizeof(w));
    block_encode_into(pcm, n, zero_run, partitioning, &w, NULL);
    *out = w.buf ? w.buf : (uint8_t*)malloc(1);
    *out_size = w.len;
    return 0;
}

int laco_block_plan(const int32_t* pcm, uint32_t n, int zero_run, int partitioning, laco_plan* plan) {
    int32_t* res = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
    analyze_block(pcm, n, zero_run, partitioning, plan, res);
    free(res);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Stereo estimate: src/codec/lac/encoder.cpp:31-57, 114-197
 * ---------------------------------------------------------------------------------------------- */
static uint64_t add_sat(uint64_t a, uint64_t b) { return (b > UINT64_MAX - a) ? UINT64_MAX : a + b; }
static uint64_t zz64(int64_t v) { /* lac/encoder.cpp:38-41 */
    return v >= 0 ? ((uint64_t)v << 1) : ((((uint64_t)(-(v + 1))) << 1) | 1u);
}
static uint64_t approx_rice_bits(uint64_t sum, uint64_t count) { /* lac/encoder.cpp:43-57 */
    if (count == 0) return 0;
    const uint64_t mean = (sum + (count >> 1)) / count;
    uint32_t k = 0;
    while (k < 31u && ((uint64_t)1 << k) < mean) ++k;
    return add_sat(sum >> k, count * (uint64_t)(k + 1u));
}

void laco_stereo_estimate(const int32_t* left, const int32_t* right, uint32_t n, laco_stereo* out) {
    uint64_t s[12];
    memset(s, 0, sizeof(s));
    int64_t pl = 0, pr = 0, pm = 0, ps = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const int64_t l = left[i], r = right[i];
        const int64_t m = (l + r) >> 1, sd = l - r;
        const int64_t cur[4] = {l, r, m, sd};
        const int64_t prev[4] = {pl, pr, pm, ps};
        for (int c = 0; c < 4; ++c) {
            s[c] = add_sat(s[c], zz64(cur[c]));
            if (i == 0) {
                s[4 + c] = zz64(cur[c]);
                s[8 + c] = s[4 + c];
            } else {
                s[4 + c] = add_sat(s[4 + c], zz64(cur[c] - prev[c]));
                s[8 + c] = add_sat(s[8 + c], zz64(cur[c] + prev[c]));
            }
        }
        pl = l;
        pr = r;
        pm = m;
        ps = sd;
    }
    uint64_t bits[4];
    int active = 0;
    for (int c = 0; c < 4; ++c) { /* lac/encoder.cpp:114-124 */
        const uint64_t raw = approx_rice_bits(s[c], n);
        const uint64_t dif = approx_rice_bits(s[4 + c], n);
        const uint64_t ant = approx_rice_bits(s[8 + c], n);
        uint64_t mn = raw < dif ? raw : dif;
        if (ant < mn) mn = ant;
        bits[c] = mn;
        active |= (raw < dif) || (ant < dif);
    }
    const uint64_t lr = add_sat(bits[0], bits[1]);
    const uint64_t ms = add_sat(bits[2], bits[3]);
    const uint64_t smaller = lr < ms ? lr : ms;
    const uint64_t diff = lr >= ms ? lr - ms : ms - lr;
    out->choose_ms = ms < lr;
    out->uncertain = smaller == 0 || diff == 0 || active || diff <= smaller / 100u;
    memcpy(out->sums, s, sizeof(s));
}

/* ------------------------------------------------------------------------------------------------
 * Stream encoder: src/codec/lac/encoder.cpp:215-466
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const int32_t *left, *right;
    uint64_t frames;
    int stereo_mode, zero_run, partitioning, channels;
    uint32_t nblocks;
    uint8_t** payload;
    size_t* payload_len;
    pthread_mutex_t mu;
    uint32_t next;
} stream_job;

static size_t encode_pair(const stream_job* j, uint64_t start, uint32_t size, int ms, bitw* w) {
    /* encode_lr / encode_ms: lac/encoder.cpp:284-317 ; M/S: simd/neon.cpp:14-30 */
    const size_t before = w->len;
    if (!ms) {
        block_encode_into(j->left + start, size, j->zero_run, j->partitioning, w, NULL);
        if (j->channels == 2) block_encode_into(j->right + start, size, j->zero_run, j->partitioning, w, NULL);
    } else {
        int32_t* m = (int32_t*)malloc(sizeof(int32_t) * size);
        int32_t* s = (int32_t*)malloc(sizeof(int32_t) * size);
        for (uint32_t i = 0; i < size; ++i) {
            const int32_t l = j->left[start + i], r = j->right[start + i];
            m[i] = (int32_t)((uint32_t)l + (uint32_t)r) >> 1;
            s[i] = (int32_t)((uint32_t)l - (uint32_t)r);
        }
        block_encode_into(m, size, j->zero_run, j->partitioning, w, NULL);
        block_encode_into(s, size, j->zero_run, j->partitioning, w, NULL);
        free(m);
        free(s);
    }
    return w->len - before;
}

static void encode_stream_block(stream_job* j, uint32_t bi) { /* lac/encoder.cpp:270-383 */
    const uint64_t start = (uint64_t)bi * kMaxBlock;
    const uint64_t rem = j->frames - start;
    const uint32_t size = rem < (uint64_t)kMaxBlock ? (uint32_t)rem : (uint32_t)kMaxBlock;
    bitw w;
    memset(&w, 0, sizeof(w));
    if (j->channels == 1) {
        encode_pair(j, start, size, 0, &w);
    } else if (j->stereo_mode == 1) {
        encode_pair(j, start, size, 1, &w);
    } else if (j->stereo_mode == 0) {
        encode_pair(j, start, size, 0, &w);
    } else {
        laco_stereo d;
        laco_stereo_estimate(j->left + start, j->right + start, size, &d);
        int choose_ms = d.choose_ms;
        if (d.uncertain) {
            if (size <= 4096u) { /* lac/encoder.cpp:336-340 */
                bitw a, b;
                memset(&a, 0, sizeof(a));
                memset(&b, 0, sizeof(b));
                const size_t lr = encode_pair(j, start, size, 0, &a);
                const size_t ms = encode_pair(j, start, size, 1, &b);
                choose_ms = ms < lr;
                free(a.buf);
                free(b.buf);
            } else { /* lac/encoder.cpp:341-354 */
                const uint64_t ps[3] = {start, start + (size - 256u) / 2u, start + size - 256u};
                size_t lr = 0, ms = 0;
                for (int p = 0; p < 3; ++p) {
                    bitw a, b;
                    memset(&a, 0, sizeof(a));
                    memset(&b, 0, sizeof(b));
                    lr += encode_pair(j, ps[p], 256u, 0, &a);
                    ms += encode_pair(j, ps[p], 256u, 1, &b);
                    free(a.buf);
                    free(b.buf);
                }
                choose_ms = ms < lr;
            }
        }
        bw_push(&w, (uint8_t)(choose_ms ? 1 : 0)); /* lac/encoder.cpp:363 */
        encode_pair(j, start, size, choose_ms, &w);
    }
    j->payload[bi] = w.buf;
    j->payload_len[bi] = w.len;
}

static void* stream_worker(void* arg) { /* lac/encoder.cpp:404-435 */
    stream_job* j = (stream_job*)arg;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        const uint32_t bi = j->next < j->nblocks ? j->next++ : UINT32_MAX;
        pthread_mutex_unlock(&j->mu);
        if (bi == UINT32_MAX) return NULL;
        encode_stream_block(j, bi);
    }
}

int laco_encode(const int32_t* left, const int32_t* right, uint64_t frames, uint32_t sample_rate,
                int bit_depth, int stereo_mode, int zero_run, int partitioning, int threads,
                uint8_t** out, uint64_t* out_size) {
    /* validation: lac/encoder.cpp:220-241, 71-102 */
    if (frames == 0 || left == NULL) return 1;
    if (!(sample_rate == 44100 || sample_rate == 48000 || sample_rate == 96000 || sample_rate == 192000))
        return 1;
    if (!(bit_depth == 16 || bit_depth == 24)) return 1;
    if (stereo_mode < 0 || stereo_mode > 2) return 1;
    const int32_t lo = bit_depth == 16 ? -32768 : -0x800000;
    const int32_t hi = bit_depth == 16 ? 32767 : 0x7FFFFF;
    for (uint64_t i = 0; i < frames; ++i) {
        if (left[i] < lo || left[i] > hi) return 1;
        if (right && (right[i] < lo || right[i] > hi)) return 1;
    }
    stream_job j;
    memset(&j, 0, sizeof(j));
    j.left = left;
    j.right = right;
    j.frames = frames;
    j.channels = right ? 2 : 1;
    j.stereo_mode = right ? stereo_mode : 0;
    j.zero_run = zero_run;
    j.partitioning = partitioning;
    j.nblocks = (uint32_t)((frames + kMaxBlock - 1) / kMaxBlock); /* plan_blocks: lac/encoder.cpp:59-69 */
    j.payload = (uint8_t**)calloc(j.nblocks, sizeof(uint8_t*));
    j.payload_len = (size_t*)calloc(j.nblocks, sizeof(size_t));
    pthread_mutex_init(&j.mu, NULL);
    int nt = threads > 0 ? threads : 1;
    if ((uint32_t)nt > j.nblocks) nt = (int)j.nblocks;
    if (nt <= 1) {
        stream_worker(&j);
    } else {
        pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nt);
        for (int t = 0; t < nt; ++t) pthread_create(&th[t], NULL, stream_worker, &j);
        for (int t = 0; t < nt; ++t) pthread_join(th[t], NULL);
        free(th);
    }
    pthread_mutex_destroy(&j.mu);

    /* container: frame_header.hpp:25-36, lac/encoder.cpp:243-250, 445-465 */
    size_t total = 10 + 4 + 8 * (size_t)j.nblocks;
    for (uint32_t b = 0; b < j.nblocks; ++b) total += j.payload_len[b];
    uint8_t* o = (uint8_t*)malloc(total);
    size_t p = 0;
    o[p++] = 0x4C;
    o[p++] = 0x41;
    o[p++] = 3;
    o[p++] = (uint8_t)j.channels;
    o[p++] = (uint8_t)j.stereo_mode;
    o[p++] = (uint8_t)((sample_rate >> 8) & 0xFF);
    o[p++] = (uint8_t)(sample_rate & 0xFF);
    o[p++] = (uint8_t)((sample_rate >> 16) & 0xFF);
    o[p++] = (uint8_t)bit_depth;
    o[p++] = 0;
#define PUT32(v)                         \
    do {                                 \
        const uint32_t vv = (uint32_t)(v); \
        o[p++] = (uint8_t)(vv >> 24);    \
        o[p++] = (uint8_t)(vv >> 16);    \
        o[p++] = (uint8_t)(vv >> 8);     \
        o[p++] = (uint8_t)vv;            \
    } while (0)
    PUT32(j.nblocks);
    int bad = 0;
    for (uint32_t b = 0; b < j.nblocks; ++b) {
        const uint64_t start = (uint64_t)b * kMaxBlock;
        const uint64_t rem = frames - start;
        PUT32(rem < (uint64_t)kMaxBlock ? rem : (uint64_t)kMaxBlock);
        PUT32(j.payload_len[b]);
        if (j.payload_len[b] == 0 || j.payload_len[b] > UINT32_MAX) bad = 1;
    }
#undef PUT32
    for (uint32_t b = 0; b < j.nblocks; ++b) {
        memcpy(o + p, j.payload[b], j.payload_len[b]);
        p += j.payload_len[b];
        free(j.payload[b]);
    }
    free(j.payload);
    free(j.payload_len);
    if (bad) {
        free(o);
        return 2;
    }
    *out = o;
    *out_size = total;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Decoder (v3 streams): src/codec/block/decoder.cpp:64-520, src/codec/lac/decoder.cpp:48-65,76-303,
 * src/codec/bitstream/bit_reader.hpp
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const uint8_t* d;
    uint64_t nbits, pos;
    int err;
} bitr;

static uint32_t br_bit(bitr* r) {
    if (r->pos >= r->nbits) {
        r->err = 1;
        return 0;
    }
    const uint32_t b = (r->d[r->pos >> 3] >> (7 - (r->pos & 7))) & 1u;
    ++r->pos;
    return b;
}
static uint32_t br_bits(bitr* r, int n) {
    uint32_t v = 0;
    for (int i = 0; i < n; ++i) v = (v << 1) | br_bit(r);
    return v;
}
static int br_unary(bitr* r, uint32_t max_q, uint32_t* q) {
    uint32_t c = 0;
    for (;;) {
        const uint32_t b = br_bit(r);
        if (r->err) return 0;
        if (!b) break;
        if (c == max_q) return 0;
        ++c;
    }
    *q = c;
    return 1;
}
static int read_rice_unsigned(bitr* r, uint32_t k, uint32_t* value) { /* block/decoder.cpp:76-86 */
    if (k > 31u) return 0;
    uint32_t q = 0;
    if (!br_unary(r, UINT32_MAX >> k, &q)) return 0;
    const uint32_t rem = k ? br_bits(r, (int)k) : 0u;
    if (r->err) return 0;
    *value = (q << k) | rem;
    return 1;
}
static int32_t unzigzag(uint32_t u) {
    return (u & 1u) ? (int32_t)(-(int64_t)((u >> 1) + 1u)) : (int32_t)(u >> 1);
}

static int decode_segment(bitr* r, uint32_t samples, uint32_t k0, uint8_t mode, int32_t* res, int stateless) {
    uint32_t k = k0, count = 0;
    uint64_t sum = 0;
    adapt_state st;
    if (!stateless) adapt_init(&st);
#define ADAPT() (stateless ? adapt_stateless(sum, count) : adapt_k(sum, count, &st))
    if (mode == kModeRice) {
        for (uint32_t i = 0; i < samples; ++i) {
            uint32_t u;
            if (!read_rice_unsigned(r, k, &u)) return 0;
            res[i] = unzigzag(u);
            sum += u;
            ++count;
            k = ADAPT();
        }
        return 1;
    }
    if (mode == kModeStatic) {
        for (uint32_t i = 0; i < samples; ++i) {
            uint32_t u;
            if (!read_rice_unsigned(r, k0, &u)) return 0;
            res[i] = unzigzag(u);
        }
        return 1;
    }
    if (mode == kModeBin) {
        for (uint32_t i = 0; i < samples; ++i) {
            const uint32_t tag = br_bits(r, 2);
            if (r->err) return 0;
            int32_t v = 0;
            uint32_t u;
            if (tag == 1 || tag == 2) {
                const uint32_t sgn = br_bit(r);
                if (r->err) return 0;
                v = (int32_t)tag * (sgn ? -1 : 1);
                u = zigzag(v);
            } else if (tag == 3) {
                if (!read_rice_unsigned(r, k, &u)) return 0;
                v = unzigzag(u);
            } else {
                u = 0;
            }
            res[i] = v;
            sum += u;
            ++count;
            k = ADAPT();
        }
        return 1;
    }
    /* zero-run */
    uint32_t idx = 0;
    while (idx < samples) {
        const uint32_t tag = br_bits(r, 2);
        if (r->err || tag > 2) return 0;
        if (tag == 0) {
            uint32_t u;
            if (!read_rice_unsigned(r, k, &u)) return 0;
            res[idx++] = unzigzag(u);
            sum += u;
            ++count;
            k = ADAPT();
        } else if (tag == 1) {
            uint32_t enc;
            if (!read_rice_unsigned(r, kZeroRunK, &enc)) return 0;
            const uint64_t run = (uint64_t)enc + kZeroRunMin;
            if (run > samples - idx) return 0;
            for (uint32_t j = 0; j < run; ++j) res[idx + j] = 0;
            idx += (uint32_t)run;
            if (stateless) {
                count += (uint32_t)run;
                k = adapt_stateless(sum, count);
            } else {
                for (uint32_t j = 0; j < run; ++j) {
                    ++count;
                    k = adapt_k(sum, count, &st);
                }
            }
        } else {
            const uint32_t zz = br_bits(r, 32);
            if (r->err) return 0;
            const int32_t v = unzigzag(zz);
            res[idx++] = v;
            sum += zigzag(v);
            ++count;
            k = ADAPT();
        }
    }
#undef ADAPT
    return idx == samples;
}

static int decode_channel_block(bitr* r, uint32_t n, int32_t* out) { /* block/decoder.cpp:64-520 */
    const uint8_t type = (uint8_t)br_bits(r, 8);
    const int order = (int)br_bits(r, 8);
    if (r->err || type > 2) return 0;
    if (type == 2) {
        if (order <= 0 || order > 32 || (uint32_t)order >= n) return 0;
    } else if (type == 1) {
        if (order != 2) return 0;
    } else if (order > 4) {
        return 0;
    }
    int16_t c[33];
    memset(c, 0, sizeof(c));
    if (type == 2)
        for (int i = 1; i <= order; ++i) c[i] = (int16_t)br_bits(r, 16);
    const uint8_t control = (uint8_t)br_bits(r, 8);
    if (r->err || (control & 0x10u)) return 0;
    const int pflag = (control & 0x80u) != 0;
    const uint8_t p = control & 0x0Fu;
    const uint8_t cmode = (control >> 5) & 3u;
    if ((pflag && p == 0) || (!pflag && p != 0) || p > kMaxPartitionOrder) return 0;
    if (p > 0 && (n >> p) < (uint32_t)kMinPartition) return 0;
    uint32_t sizes[LACO_MAX_PARTS];
    const uint32_t parts = partition_sizes(n, p, sizes);
    uint8_t modes[LACO_MAX_PARTS], ks[LACO_MAX_PARTS];
    for (uint32_t i = 0; i < parts; ++i) {
        modes[i] = (uint8_t)br_bits(r, 2);
        ks[i] = (uint8_t)br_bits(r, 5);
        if (r->err) return 0;
    }
    if (modes[0] != cmode) return 0;
    uint32_t off = 0;
    for (uint32_t i = 0; i < parts; ++i) {
        if (!decode_segment(r, sizes[i], ks[i], modes[i], out + off, p > 0)) return 0;
        off += sizes[i];
    }
    /* byte alignment with zero padding: bit_reader.hpp consume_zero_padding_to_byte */
    while (r->pos & 7u) {
        if (br_bit(r)) return 0;
    }
    /* reconstruction in place */
    if (type == 0) {
        for (uint32_t i = (uint32_t)order; i < n; ++i) {
            int64_t pred = 0;
            switch (order) {
                case 1: pred = out[i - 1]; break;
                case 2: pred = 2LL * out[i - 1] - out[i - 2]; break;
                case 3: pred = 3LL * out[i - 1] - 3LL * out[i - 2] + out[i - 3]; break;
                case 4: pred = 4LL * out[i - 1] - 6LL * out[i - 2] + 4LL * out[i - 3] - out[i - 4]; break;
                default: break;
            }
            const int64_t s = (int64_t)out[i] + pred;
            if (s < INT32_MIN || s > INT32_MAX) return 0;
            out[i] = (int32_t)s;
        }
    } else if (type == 1) {
        for (uint32_t i = 2; i < n; ++i) {
            const int64_t pred = (3LL * out[i - 1] - (int64_t)out[i - 2]) >> 2;
            const int64_t s = (int64_t)out[i] + pred;
            if (s < INT32_MIN || s > INT32_MAX) return 0;
            out[i] = (int32_t)s;
        }
    } else {
        for (uint32_t i = 0; i < n; ++i) {
            int64_t acc = 0;
            const int taps = order < (int)i ? order : (int)i;
            for (int t = 1; t <= taps; ++t) acc += (int64_t)c[t] * (int64_t)out[i - (uint32_t)t];
            const int64_t s = (acc >> 15) + (int64_t)out[i];
            if (s < INT32_MIN || s > INT32_MAX) return 0;
            out[i] = (int32_t)s;
        }
    }
    return 1;
}

int laco_decode(const uint8_t* data, uint64_t size, int32_t** left, int32_t** right, uint64_t* frames,
                int* channels, uint32_t* sample_rate, int* bit_depth, int* stereo_mode) {
    if (size < 14 || data[0] != 0x4C || data[1] != 0x41 || data[2] != 3) return 1;
    const int ch = data[3], sm = data[4];
    const uint32_t sr = ((uint32_t)data[5] << 8) | data[6] | ((uint32_t)data[7] << 16);
    const int bd = data[8];
    if ((ch != 1 && ch != 2) || sm > 2 || (bd != 16 && bd != 24) || data[9] != 0) return 1;
    const uint32_t nb = ((uint32_t)data[10] << 24) | ((uint32_t)data[11] << 16) | ((uint32_t)data[12] << 8) | data[13];
    if (size < 14 + 8ull * nb) return 1;
    uint64_t total = 0, pay = 0;
    for (uint32_t b = 0; b < nb; ++b) {
        const uint8_t* e = data + 14 + 8ull * b;
        total += ((uint32_t)e[0] << 24) | ((uint32_t)e[1] << 16) | ((uint32_t)e[2] << 8) | e[3];
        pay += ((uint32_t)e[4] << 24) | ((uint32_t)e[5] << 16) | ((uint32_t)e[6] << 8) | e[7];
    }
    if (14 + 8ull * nb + pay != size) return 1;
    int32_t* L = (int32_t*)malloc(sizeof(int32_t) * (total ? total : 1));
    int32_t* R = ch == 2 ? (int32_t*)malloc(sizeof(int32_t) * (total ? total : 1)) : NULL;
    uint64_t off = 0, poff = 14 + 8ull * nb;
    const int32_t lo = bd == 16 ? -32768 : -0x800000, hi = bd == 16 ? 32767 : 0x7FFFFF;
    int ok = 1;
    for (uint32_t b = 0; b < nb && ok; ++b) {
        const uint8_t* e = data + 14 + 8ull * b;
        const uint32_t n = ((uint32_t)e[0] << 24) | ((uint32_t)e[1] << 16) | ((uint32_t)e[2] << 8) | e[3];
        const uint32_t by = ((uint32_t)e[4] << 24) | ((uint32_t)e[5] << 16) | ((uint32_t)e[6] << 8) | e[7];
        bitr r = {data + poff, 8ull * by, 0, 0};
        int ms = (sm == 1);
        if (n == 0 || n > (uint32_t)kMaxBlock) ok = 0;
        if (ok && ch == 2 && sm == 2) {
            const uint32_t flag = br_bits(&r, 8);
            if (flag > 1) ok = 0;
            ms = (int)flag;
        }
        if (ok) ok = decode_channel_block(&r, n, L + off);
        if (ok && ch == 2) ok = decode_channel_block(&r, n, R + off);
        if (ok && r.pos != r.nbits) ok = 0;
        if (ok && ch == 2 && ms) { /* lac/decoder.cpp:48-65 */
            for (uint32_t i = 0; i < n; ++i) {
                const int64_t m = L[off + i], s = R[off + i];
                const int64_t l = m + ((s + (s & 1)) >> 1);
                const int64_t rr = l - s;
                L[off + i] = (int32_t)l;
                R[off + i] = (int32_t)rr;
            }
        }
        if (ok) {
            for (uint32_t i = 0; i < n; ++i) {
                if (L[off + i] < lo || L[off + i] > hi) ok = 0;
                if (R && (R[off + i] < lo || R[off + i] > hi)) ok = 0;
            }
        }
        off += n;
        poff += by;
    }
    if (!ok) {
        free(L);
        free(R);
        return 1;
    }
    *left = L;
    *right = R;
    *frames = total;
    *channels = ch;
    *sample_rate = sr;
    *bit_depth = bd;
    *stereo_mode = sm;
    return 0;
}
