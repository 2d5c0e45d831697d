// wide.hip -- Block::Encoder::encode for blocks outside the 25-bit domain of the streaming kernels.
//
// LAC::Encoder validates its input to 16 / 24 bits, so the streaming kernels (k_front / k_analyze / k_emit) only ever see |x| <= 2^24 (mid/side
// included): residuals below 2^30, 32-bit fast paths, two flag bits inside the residual words.  Block::Encoder itself
// has no such limit (ref src/codec/block/encoder.cpp:313-316 takes any int32 samples and never throws): there the LPC
// residual can leave int32 and the encoder falls back to the next lower order of {12, 10, 8, 6, 4}, finally to order 0
// (ref src/codec/lpc/lpc.cpp:24-36, 188-229), zigzag values use all 32 bits and the Rice parameter saturates at 31.
// That domain gets this kernel: one workgroup per block, the reference's own serial formulation with the parallelism
// the data offers -- samples for residuals and static costs, one lane per candidate for the stateful adaptive model
// (its state is the reference's: ring of the last 256 magnitudes, two 96-entry flag rings, a 64-bit division per
// sample), one lane per partition for the partition search.  It is a correctness path, not a fast one (tens of
// milliseconds per block); nothing of the streaming pipeline goes through it.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "x87.h"

namespace lacx {

namespace {

constexpr int kWideThreads = 512;
constexpr uint32_t kInitialScan = 256;  // ref block/encoder.cpp:121-158 (kInitialScanCount), candidates k = 0..12
constexpr uint32_t kZeroRunMinW = 4, kZeroRunKW = 2;

__device__ __forceinline__ uint32_t zz(int32_t r) { return ((uint32_t)r << 1) ^ (r < 0 ? 0xFFFFFFFFu : 0u); }
__device__ __forceinline__ uint64_t rice_bits_w(uint32_t u, uint32_t k) {  // ref block/encoder.cpp:67-70
    return (uint64_t)(k >= 31u ? 0u : (u >> k)) + 1u + k;
}
__device__ __forceinline__ uint32_t bit_width_u64(uint64_t v) { return v ? 64u - (uint32_t)__clzll((long long)v) : 0u; }
__device__ __forceinline__ uint32_t adapt_stateless_w(uint64_t sum, uint32_t count) {  // ref block/encoder.cpp:72-77
    if (count == 0) return 0;
    const uint64_t mean = (sum + (count >> 1)) / count;
    if (mean <= 1) return 0;
    const uint32_t w = bit_width_u64(mean - 1u);
    return w < 31u ? w : 31u;
}

// Rice::AdaptState / Rice::adapt_k (ref src/codec/rice/rice.hpp:15-32, 45-114); rings in LDS, one instance per lane.
struct AdaptW {
    uint64_t previous_sum, window_sum;
    uint32_t window_index, micro_index, window_filled, large_q, zero_q;
    uint32_t* recent_u;    // [256]
    uint8_t* large_flags;  // [96]
    uint8_t* zero_flags;   // [96]
};
__device__ void adapt_init_w(AdaptW& s) {
    s.previous_sum = s.window_sum = 0;
    s.window_index = s.micro_index = s.window_filled = s.large_q = s.zero_q = 0;
    for (int i = 0; i < 256; ++i) s.recent_u[i] = 0;
    for (int i = 0; i < 96; ++i) s.large_flags[i] = s.zero_flags[i] = 0;
}
__device__ uint32_t adapt_k_w(uint64_t sum, uint32_t count, AdaptW& s) {
    if (count == 0) return 0;
    const uint64_t cur = sum - s.previous_sum;
    s.previous_sum = sum;
    const uint32_t mi = s.micro_index;
    s.large_q -= s.large_flags[mi];
    s.zero_q -= s.zero_flags[mi];
    if (s.window_filled < 256u) {
        ++s.window_filled;
    } else {
        s.window_sum -= s.recent_u[s.window_index];
    }
    s.recent_u[s.window_index] = (uint32_t)cur;
    s.window_sum += cur;
    const uint64_t mean = (sum + (count >> 1)) / count;
    uint32_t k = 0;
    if (mean > 1) {
        k = bit_width_u64(mean - 1u);
        if (k > 31u) k = 31u;
    }
    const uint32_t q_base = k >= 31u ? 0u : (uint32_t)(cur >> k);
    const uint8_t is_large = q_base > 3u, is_zero = q_base == 0u;
    s.large_q += is_large;
    s.zero_q += is_zero;
    s.large_flags[mi] = is_large;
    s.zero_flags[mi] = is_zero;
    int bias = 0;
    if (s.window_filled > 0 && mean > 0) {
        const uint64_t local_mean = s.window_filled == 256u ? ((s.window_sum + 128u) >> 8)
                                                             : ((s.window_sum + (s.window_filled >> 1)) / s.window_filled);
        if (local_mean * 3 > mean * 4) {
            bias = 1;
        } else if (local_mean * 4 + 3 < mean * 3) {
            bias = -1;
        }
    }
    if (s.window_index + 1 >= 96u || s.window_filled >= 96u) {
        const uint32_t ws = s.window_filled >= 96u ? 96u : s.window_filled;
        if (s.large_q * 4 >= ws * 3) {
            bias = bias + 1 < 1 ? bias + 1 : 1;
        } else if (s.zero_q * 5 >= ws * 4) {
            bias = bias - 1 > -1 ? bias - 1 : -1;
        }
    }
    int bk = (int)k + bias;
    if (bk < 0) bk = 0;
    if (bk > 31) bk = 31;
    s.micro_index = s.micro_index + 1u == 96u ? 0u : s.micro_index + 1u;
    s.window_index = (s.window_index + 1u) & 255u;
    return (uint32_t)bk;
}

struct CostsW {
    uint64_t rice, zr, bin;
    uint32_t has_run;
};
// estimate_residual_costs (ref block/encoder.cpp:201-263): one lane walks the segment.
__device__ CostsW estimate_costs_w(const int32_t* r, uint32_t n, uint32_t initial_k, AdaptW* st /* null: stateless */) {
    CostsW c{0, 0, 0, 0};
    uint32_t k = initial_k, count = 0, idx = 0;
    uint64_t sum = 0;
    while (idx < n) {
        uint32_t run = 0;
        while (idx + run < n && r[idx + run] == 0) ++run;
        if (run >= kZeroRunMinW) {
            c.has_run = 1;
            c.zr += 2 + rice_bits_w(run - kZeroRunMinW, kZeroRunKW);
            for (uint32_t j = 0; j < run; ++j) {
                c.rice += rice_bits_w(0, k);
                c.bin += 2;
                ++count;
                k = st ? adapt_k_w(sum, count, *st) : adapt_stateless_w(sum, count);
            }
            idx += run;
            continue;
        }
        const int32_t v = r[idx];
        const uint32_t u = zz(v);
        const uint64_t rb = rice_bits_w(u, k);
        c.rice += rb;
        c.bin += v == 0 ? 2u : ((v == 1 || v == -1 || v == 2 || v == -2) ? 3u : 2u + rb);
        const uint32_t esc = 1u << ((k + 3u) < 24u ? (k + 3u) : 24u);
        c.zr += 2 + (u > esc ? 32u : rb);
        sum += u;
        ++count;
        k = st ? adapt_k_w(sum, count, *st) : adapt_stateless_w(sum, count);
        ++idx;
    }
    return c;
}

// estimate_initial_k (ref :121-158): lowest-cost k in 0..12 over the first min(256, n) samples, lowest k on a tie.
__device__ uint32_t initial_k_w(const int32_t* r, uint32_t n) {
    if (n == 0) return 0;
    const uint32_t count = n < kInitialScan ? n : kInitialScan;
    uint64_t cost[13];
    for (int k = 0; k <= 12; ++k) cost[k] = 0;
    for (uint32_t i = 0; i < count; ++i) {
        const uint32_t u = zz(r[i]);
        for (uint32_t k = 0; k <= 12u; ++k) cost[k] += (uint64_t)(u >> k) + 1u + k;
    }
    uint32_t best_k = 0;
    uint64_t best = ~0ull;
    for (uint32_t k = 0; k <= 12u; ++k) {
        if (cost[k] < best) {
            best = cost[k];
            best_k = k;
        }
    }
    return best_k;
}
// estimate_static_k + estimate_static_rice_bits (ref :160-188)
__device__ uint32_t static_k_w(const int32_t* r, uint32_t n, uint64_t* bits) {
    uint64_t cost[16];
    for (int k = 0; k < 16; ++k) cost[k] = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t u = zz(r[i]);
        for (uint32_t k = 0; k < 16u; ++k) cost[k] += rice_bits_w(u, k);
    }
    uint32_t best_k = 0;
    uint64_t best = ~0ull;
    for (uint32_t k = 0; k < 16u; ++k) {
        if (cost[k] < best) {
            best = cost[k];
            best_k = k;
        }
    }
    *bits = n ? best : 0;
    return n ? best_k : 0;
}

struct EvalW {
    uint64_t rice, zr, bin, stat, best;
    uint32_t initial_k, static_k, has_run, valid;
    int used_order;
};

struct WideShared {
    unsigned long long acorr[13];
    int16_t coef[5][13];
    uint8_t used[5];
    EvalW ev[11];
    int best_cand;
    uint32_t overflow;  // an LPC residual left int32 in the current attempt
    // partition search
    uint64_t seg_bits[510];
    uint8_t seg_mode_k[510];
    // rings of the stateful model, one set per candidate lane
    uint32_t recent[11][256];
    uint8_t lflag[11][96], zflag[11][96];
};

}  // namespace

// One workgroup, one block.  res: scratch for the eleven candidate residuals, [11][kMaxBlock] int32.
__global__ __launch_bounds__(kWideThreads) void k_wide_block(const int32_t* __restrict__ x, uint32_t n, int zero_run,
                                                              int partitioning, int32_t* __restrict__ res,
                                                              ChannelPlan* __restrict__ out) {
    __shared__ WideShared sh;
    const int tid = threadIdx.x;
    const int max_valid_order = n > 1 ? (int)(n - 1 < 32u ? n - 1 : 32u) : 0;
    // ---- exact autocorrelation, 13 lags (ref lpc.cpp:80-96; int64, wrapping like the reference's accumulation) ------
    if (tid < 13) sh.acorr[tid] = 0;
    if (tid == 0) sh.best_cand = -1;
    __syncthreads();
    {
        unsigned long long acc[13];
        for (int k = 0; k < 13; ++k) acc[k] = 0;
        for (uint32_t i = (uint32_t)tid; i < n; i += kWideThreads)
            for (uint32_t k = 0; k < 13u && k <= i; ++k) acc[k] += (unsigned long long)((long long)x[i] * (long long)x[i - k]);
        for (int k = 0; k < 13; ++k) atomicAdd(&sh.acorr[k], acc[k]);
    }
    __syncthreads();
    // ---- Levinson-Durbin in x87 extended precision (x87.h), all candidate orders from one recursion ------------------
    if (tid == 0) {
        XfLocalArray R, a, p;
        levinson_candidates_t([&](int i) { return (int64_t)sh.acorr[i]; }, max_valid_order, R, a, p,
                              [&](int ci, int j, int16_t v) { sh.coef[ci][j] = v; }, [&](int ci, uint8_t v) { sh.used[ci] = v; });
    }
    __syncthreads();
    // ---- the eleven candidate residuals (ref block/encoder.cpp:265-309, lpc.cpp:188-229) -----------------------------
    for (int cand = 0; cand <= 10; ++cand) {
        int32_t* r = res + (size_t)cand * kMaxBlock;
        if (tid == 0) {
            sh.ev[cand].valid = cand < 6 ? 1u : 0u;
            sh.ev[cand].used_order = cand < 6 ? 0 : (int)sh.used[cand - 6];
        }
        if (cand <= 4) {
            for (uint32_t i = (uint32_t)tid; i < n; i += kWideThreads) {
                long long pred = 0;
                if (i >= (uint32_t)cand) {
                    switch (cand) {
                        case 1: pred = x[i - 1]; break;
                        case 2: pred = 2LL * x[i - 1] - x[i - 2]; break;
                        case 3: pred = 3LL * x[i - 1] - 3LL * x[i - 2] + x[i - 3]; break;
                        case 4: pred = 4LL * x[i - 1] - 6LL * x[i - 2] + 4LL * x[i - 3] - x[i - 4]; break;
                        default: break;
                    }
                }
                r[i] = (int32_t)((long long)x[i] - pred);
            }
        } else if (cand == 5) {
            for (uint32_t i = (uint32_t)tid; i < n; i += kWideThreads) {
                const long long pred = i >= 2 ? ((3LL * (long long)x[i - 1] - (long long)x[i - 2]) >> 2) : 0;
                r[i] = (int32_t)((long long)x[i] - pred);
            }
        } else {
            // attempts: the achieved order, then every order of {12, 10, 8, 6, 4} below it, finally none (order 0, which
            // the encoder then skips: ref block/encoder.cpp:394-399)
            const int ci = cand - 6, cand_order = 4 + 2 * ci;
            int order = (int)sh.used[ci];  // uniform
            if (order > cand_order) order = cand_order;
            __syncthreads();
            while (order > 0) {
                if (tid == 0) sh.overflow = 0;
                __syncthreads();
                uint32_t bad = 0;
                for (uint32_t i = (uint32_t)tid; i < n; i += kWideThreads) {
                    long long acc = 0;
                    const int taps = (uint32_t)order < i ? order : (int)i;
                    for (int t = 1; t <= taps; ++t) acc += (long long)sh.coef[ci][t] * (long long)x[i - (uint32_t)t];
                    const long long diff = (long long)x[i] - (acc >> 15);
                    if (diff < -2147483648LL || diff > 2147483647LL) bad = 1;
                    r[i] = (int32_t)diff;
                }
                if (bad) atomicOr(&sh.overflow, 1u);
                __syncthreads();
                if (!sh.overflow) break;  // uniform
                int next = 0;
                for (int o = 12; o >= 4; o -= 2)
                    if (o < order && o <= cand_order) {
                        next = o;
                        break;
                    }
                order = next;
                __syncthreads();
            }
            if (tid == 0) {
                sh.ev[cand].used_order = order;
                sh.ev[cand].valid = order > 0 ? 1u : 0u;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- exact costs: one lane per candidate (ref block/encoder.cpp:337-351) ----------------------------------------
    if (tid <= 10 && sh.ev[tid].valid) {
        const int32_t* r = res + (size_t)tid * kMaxBlock;
        EvalW& ev = sh.ev[tid];
        ev.initial_k = initial_k_w(r, n);
        AdaptW st;
        st.recent_u = sh.recent[tid];
        st.large_flags = sh.lflag[tid];
        st.zero_flags = sh.zflag[tid];
        adapt_init_w(st);
        const CostsW c = estimate_costs_w(r, n, ev.initial_k, &st);
        ev.rice = c.rice;
        ev.has_run = c.has_run;
        ev.zr = (zero_run && c.has_run) ? c.zr : c.rice;
        ev.bin = c.bin;
        ev.static_k = static_k_w(r, n, &ev.stat);
        const uint64_t a = ev.rice < ev.stat ? ev.rice : ev.stat, b = ev.zr < ev.bin ? ev.zr : ev.bin;
        ev.best = a < b ? a : b;
    }
    __syncthreads();
    if (tid == 0) {  // the first candidate with the strictly smallest cost (ref :352-359)
        int best = -1;
        for (int c = 0; c <= 10; ++c)
            if (sh.ev[c].valid && (best < 0 || sh.ev[c].best < sh.ev[best].best)) best = c;
        sh.best_cand = best;
    }
    __syncthreads();
    const int best = sh.best_cand;
    const int32_t* br = res + (size_t)best * kMaxBlock;
    // ---- partition search: one lane per partition of every order (ref block/encoder.cpp:486-552) ---------------------
    int max_p = 0;
    if (partitioning && n >= (uint32_t)kMinPartition)
        for (int p = 1; p <= kMaxPartitionOrder && (n >> p) >= (uint32_t)kMinPartition; ++p) max_p = p;
    const int nseg = max_p ? (2 << max_p) - 2 : 0;
    for (int idx = tid; idx < nseg; idx += kWideThreads) {
        const int p = 31 - __clz(idx + 2);
        const uint32_t part = (uint32_t)(idx + 2 - (1 << p)), parts = 1u << p, base = n >> p;
        const uint32_t s = part * base, len = part + 1u == parts ? n - s : base;
        const int32_t* seg = br + s;
        const uint32_t ak = initial_k_w(seg, len);
        uint64_t sbits;
        const uint32_t sk = static_k_w(seg, len, &sbits);
        const CostsW c = estimate_costs_w(seg, len, ak, nullptr);
        const bool allow_zr = zero_run && c.has_run;
        uint32_t mode = 0, k = ak;
        uint64_t bits = c.rice;
        if (allow_zr && c.zr < bits) {
            mode = 1;
            bits = c.zr;
        }
        if (c.bin < bits) {
            mode = 2;
            bits = c.bin;
        }
        if (sbits < bits || sbits <= bits + bits / 20u) {
            mode = 3;
            k = sk;
            bits = sbits;
        }
        sh.seg_bits[idx] = bits;
        sh.seg_mode_k[idx] = (uint8_t)((mode << 5) | k);
    }
    __syncthreads();
    if (tid == 0) {
        const EvalW& ev = sh.ev[best];
        ChannelPlan pl;
        uint8_t* raw = reinterpret_cast<uint8_t*>(&pl);
        for (size_t i = 0; i < sizeof(ChannelPlan); ++i) raw[i] = 0;
        int order;
        if (best <= 4) {
            pl.predictor_type = 0;
            order = best;
        } else if (best == 5) {
            pl.predictor_type = 1;
            order = 2;
        } else {
            pl.predictor_type = 2;
            order = ev.used_order < max_valid_order ? ev.used_order : max_valid_order;  // ref :421-423
            if (order < 1) order = 1;
            for (int i = 0; i < 12; ++i) pl.coef[i] = sh.coef[best - 6][i + 1];
        }
        pl.order = (uint8_t)order;
        pl.valid = 1;
        // unpartitioned choice (ref :432-456)
        const bool allow_zr = zero_run && ev.has_run;
        uint32_t mode = 0, k = ev.initial_k;
        uint64_t bits = ev.rice;
        if (allow_zr && ev.zr <= bits) {
            bits = ev.zr;
            mode = 1;
        }
        if (ev.bin < bits) {
            bits = ev.bin;
            mode = 2;
        }
        if (ev.stat < bits) {
            bits = ev.stat;
            mode = 3;
            k = ev.static_k;
        }
        uint64_t best_total = bits + 8u + 7u;
        best_total += (8u - (best_total & 7u)) & 7u;
        int best_p = 0;
        for (int p = 1; p <= max_p; ++p) {
            const uint32_t parts = 1u << p, segbase = (2u << (p - 1)) - 2u;
            uint64_t sum = 0;
            for (uint32_t i = 0; i < parts; ++i) sum += sh.seg_bits[segbase + i];
            uint64_t total = sum + 8u + 7ull * parts;
            total += (8u - (total & 7u)) & 7u;
            if (total < best_total || (total <= best_total + best_total / 20u && best_p == 0)) {
                best_total = total;
                best_p = p;
            }
        }
        pl.partition_order = (uint8_t)best_p;
        pl.total_bits = best_total;
        const uint64_t bytes = (16u + (pl.predictor_type == 2 ? 16u * (uint32_t)order : 0u) + best_total) >> 3;
        pl.payload_bytes = bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)bytes;
        if (best_p == 0) {
            pl.part_mode_k[0] = (uint8_t)((mode << 5) | k);
        } else {
            const uint32_t parts = 1u << best_p, segbase = (2u << (best_p - 1)) - 2u;
            for (uint32_t i = 0; i < parts; ++i) pl.part_mode_k[i] = sh.seg_mode_k[segbase + i];
        }
        *out = pl;
    }
}

hipError_t launch_wide_block(const int32_t* d_x, uint32_t n, int zero_run, int partitioning, int32_t* d_res,
                             ChannelPlan* d_plan, hipStream_t stream) {
    hipLaunchKernelGGL(k_wide_block, dim3(1), dim3(kWideThreads), 0, stream, d_x, n, zero_run, partitioning, d_res, d_plan);
    return hipGetLastError();
}

}  // namespace lacx
