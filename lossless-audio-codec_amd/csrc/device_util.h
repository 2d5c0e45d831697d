// device_util.h -- device-side helpers shared by the kernel translation units (k_front.hip, k_analyze.hip,
// k_emit.hip): diagnostic stamps, wave-level scans and reductions on the DPP network, block scans, plane / k-sum totals,
// slot geometry, the stream look-up of a launch set, XCD-aware workgroup mapping, the hand-off records of the fused emit.
// Device code only (included after <hip/hip_runtime.h>).
#pragma once
#include <hip/hip_runtime.h>

#include "analyze_core.h"
#include "kernels.h"

namespace lacx {

// Diagnostic phase stamps (only in builds made with -DLACX_STAMPS; the production kernel has none).
#ifdef LACX_STAMPS
extern __device__ unsigned long long g_stamp_acc[40];  // (defined in k_analyze.hip)
#define STAMP(k)                                                        \
    do {                                                                \
        const unsigned long long _now = __builtin_amdgcn_s_memtime();   \
        stamp_acc[k] += _now - stamp_prev;                              \
        stamp_prev = _now;                                              \
    } while (0)
#define STAMP_PARAMS , unsigned long long* stamp_acc, unsigned long long& stamp_prev
#define STAMP_ARGS , stamp_acc, stamp_prev
#else
#define STAMP(k) do { } while (0)
#define STAMP_PARAMS
#define STAMP_ARGS
#endif
// ---------------------------------------------------------------------------------------------
// wave helpers (wave = 64 lanes)
// ---------------------------------------------------------------------------------------------
// Wave-wide inclusive scans and reductions on the DPP network (gfx9 row_shr / row_bcast), which is part of the
// VALU pipeline: six full-rate moves per scan instead of six trips through the LDS crossbar (ds_bpermute).
//   row_shr:n       lane i of a 16-lane row reads lane i-n of the same row
//   row_bcast:15/31 lane 15 (31) is broadcast to the following row (two rows)
// Lanes without a source keep `identity`.  Every lane must be active.
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t identity, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_mov64(uint64_t v) {  // identity 0
    const uint32_t lo = dpp_mov<CTRL, ROW_MASK>(0u, (uint32_t)v);
    const uint32_t hi = dpp_mov<CTRL, ROW_MASK>(0u, (uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t v) {  // inclusive prefix sum over the 64 lanes
    v += dpp_mov<kDppRowShr1, 0xF>(0u, v);
    v += dpp_mov<kDppRowShr2, 0xF>(0u, v);
    v += dpp_mov<kDppRowShr4, 0xF>(0u, v);
    v += dpp_mov<kDppRowShr8, 0xF>(0u, v);
    v += dpp_mov<kDppRowBcast15, 0xA>(0u, v);
    v += dpp_mov<kDppRowBcast31, 0xC>(0u, v);
    return v;
}
__device__ __forceinline__ uint64_t wave_scan_add_u64(uint64_t v) {
    v += dpp_mov64<kDppRowShr1, 0xF>(v);
    v += dpp_mov64<kDppRowShr2, 0xF>(v);
    v += dpp_mov64<kDppRowShr4, 0xF>(v);
    v += dpp_mov64<kDppRowShr8, 0xF>(v);
    v += dpp_mov64<kDppRowBcast15, 0xA>(v);
    v += dpp_mov64<kDppRowBcast31, 0xC>(v);
    return v;
}
__device__ __forceinline__ int32_t wave_scan_max_i32(int32_t v) {  // inclusive prefix max; identity INT32_MIN
    constexpr uint32_t kMin = 0x80000000u;
    auto mx = [](int32_t a, uint32_t b) { return a > (int32_t)b ? a : (int32_t)b; };
    v = mx(v, dpp_mov<kDppRowShr1, 0xF>(kMin, (uint32_t)v));
    v = mx(v, dpp_mov<kDppRowShr2, 0xF>(kMin, (uint32_t)v));
    v = mx(v, dpp_mov<kDppRowShr4, 0xF>(kMin, (uint32_t)v));
    v = mx(v, dpp_mov<kDppRowShr8, 0xF>(kMin, (uint32_t)v));
    v = mx(v, dpp_mov<kDppRowBcast15, 0xA>(kMin, (uint32_t)v));
    v = mx(v, dpp_mov<kDppRowBcast31, 0xC>(kMin, (uint32_t)v));
    return v;
}
__device__ __forceinline__ uint32_t wave_scan_or_u32(uint32_t v) {
    v |= dpp_mov<kDppRowShr1, 0xF>(0u, v);
    v |= dpp_mov<kDppRowShr2, 0xF>(0u, v);
    v |= dpp_mov<kDppRowShr4, 0xF>(0u, v);
    v |= dpp_mov<kDppRowShr8, 0xF>(0u, v);
    v |= dpp_mov<kDppRowBcast15, 0xA>(0u, v);
    v |= dpp_mov<kDppRowBcast31, 0xC>(0u, v);
    return v;
}
// inclusive prefix minimum of 64-bit keys over the 64 lanes (identity: all ones)
__device__ __forceinline__ uint64_t wave_scan_min_u64(uint64_t v) {
    auto step = [](uint64_t a, uint32_t lo, uint32_t hi) {
        const uint64_t b = ((uint64_t)hi << 32) | lo;
        return b < a ? b : a;
    };
#define LACX_MIN_STEP(CTRL, MASK) \
    v = step(v, dpp_mov<CTRL, MASK>(0xFFFFFFFFu, (uint32_t)v), dpp_mov<CTRL, MASK>(0xFFFFFFFFu, (uint32_t)(v >> 32)))
    LACX_MIN_STEP(kDppRowShr1, 0xF);
    LACX_MIN_STEP(kDppRowShr2, 0xF);
    LACX_MIN_STEP(kDppRowShr4, 0xF);
    LACX_MIN_STEP(kDppRowShr8, 0xF);
    LACX_MIN_STEP(kDppRowBcast15, 0xA);
    LACX_MIN_STEP(kDppRowBcast31, 0xC);
#undef LACX_MIN_STEP
    return v;
}
// value of lane 63 in every lane (a scalar register)
__device__ __forceinline__ uint32_t wave_last_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ uint64_t wave_last_u64(uint64_t v) {
    return ((uint64_t)wave_last_u32((uint32_t)(v >> 32)) << 32) | wave_last_u32((uint32_t)v);
}
// reductions: the total, in every lane
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) { return wave_last_u64(wave_scan_add_u64(v)); }
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return wave_last_u32(wave_scan_add_u32(v)); }
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) { return wave_last_u32(wave_scan_or_u32(v)); }

// Sum of v over the segment of (1 << LOG) consecutive lanes that contains the lane; valid in the segment's LAST lane
// (segments are aligned: lanes [k << LOG, (k + 1) << LOG)).  The first LOG steps of the wave scan.
// LDS atomics of many lanes on ONE address are serialised lane by lane: 1024 threads adding their partial sums to the
// handful of accumulators of a low partition order cost more than the arithmetic that produced the sums -- hence one
// atomic per segment instead of one per lane wherever the lanes that share an accumulator are neighbours.
template <int LOG>
__device__ __forceinline__ uint32_t seg_sum_u32(uint32_t v) {
    if (LOG >= 1) v += dpp_mov<kDppRowShr1, 0xF>(0u, v);
    if (LOG >= 2) v += dpp_mov<kDppRowShr2, 0xF>(0u, v);
    if (LOG >= 3) v += dpp_mov<kDppRowShr4, 0xF>(0u, v);
    if (LOG >= 4) v += dpp_mov<kDppRowShr8, 0xF>(0u, v);
    if (LOG >= 5) v += dpp_mov<kDppRowBcast15, 0xA>(0u, v);
    if (LOG >= 6) v += dpp_mov<kDppRowBcast31, 0xC>(0u, v);
    return v;
}
__device__ __forceinline__ uint32_t seg_sum_u32(uint32_t v, int log2_lanes) {  // log2_lanes wave-uniform, 1..6
    switch (log2_lanes) {
        case 1: return seg_sum_u32<1>(v);
        case 2: return seg_sum_u32<2>(v);
        case 3: return seg_sum_u32<3>(v);
        case 4: return seg_sum_u32<4>(v);
        case 5: return seg_sum_u32<5>(v);
        default: return seg_sum_u32<6>(v);
    }
}

// Block exclusive scans of the per-thread values the phases left in tabP/tabNZ (sum / max).
// part 1 before the barrier, part 2 after it.
template <class G>
struct ScanRegs {
    uint64_t v, inc;
    int32_t z, zinc;
};

template <class G, class M>
__device__ __forceinline__ void scan_pz_part1(M& sh, int tid, ScanRegs<G>& r) {
    const int lane = tid & 63, wave = tid >> 6;
    r.v = sh.tabP[tid];
    r.z = sh.tabNZ[tid];
    const uint64_t inc = wave_scan_add_u64(r.v);
    const int32_t zinc = wave_scan_max_i32(r.z);
    r.inc = inc;
    r.zinc = zinc;
    if (lane == 63) {
        sh.wtotP[wave] = inc;
        sh.wtotZ[wave] = zinc;
    }
}

// Returns the block total of the summed values.
template <class G, class M>
__device__ __forceinline__ uint64_t scan_pz_part2(M& sh, int tid, const ScanRegs<G>& r) {
    const int lane = tid & 63, wave = tid >> 6;
    uint64_t base = 0, total = 0;
    int32_t zbase = -1;
#pragma unroll
    for (int w = 0; w < G::T / 64; ++w) {
        const uint64_t pw = sh.wtotP[w];
        const int32_t z = sh.wtotZ[w];
        total += pw;
        if (w < wave) {
            base += pw;
            zbase = z > zbase ? z : zbase;
        }
    }
    int32_t zprev = __shfl_up(r.zinc, 1, 64);
    if (lane == 0) zprev = -1;
    sh.tabP[tid] = base + r.inc - r.v;
    sh.tabNZ[tid] = zprev > zbase ? zprev : zbase;
    if (tid == G::T - 1) {
        sh.tabP[G::T] = base + r.inc;
        sh.tabNZ[G::T] = r.zinc > zbase ? r.zinc : zbase;
    }
    return total;
}

// Per-plane population counts of the wave's bit-sliced chunk counters, via ballots.  Ballot masks and
// their popcounts are wave-uniform, so the per-plane totals accumulate on the scalar unit; planes above
// the highest set bit in the wave are skipped.  Lanes 0..29 then add their plane's count to the block totals.
// (Measured against an all-vector form -- two planes per word, unpacked per lane and summed on the DPP network: that
// one took a quarter longer; here the scalar unit is not the bottleneck, unlike in pass 1.)
// first_plane (wave-uniform): planes below it are not needed for the block totals (the static parameters that can still
// win are all >= first_plane, static_k_candidates64); wave 0 still counts them for the first 256 samples, whose
// initial-k estimate looks at every k <= 12.
template <class G>
__device__ __forceinline__ void plane_totals_wave(const Thread<G>& th, uint32_t* planeTot, uint32_t* planeTot256,
                                                  int tid, int first_plane = 0) {
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t any = 0;
#pragma unroll
    for (int l = 0; l < G::LV; ++l) any |= th.cs[l];
    const int nplanes = 32 - __clz((int)wave_or_u32(any));  // uniform, 0..30
    // lanes of wave 0 whose chunk lies inside the first 256 samples
    const uint64_t m256 = (G::W256 >= 64) ? ~0ull : ((1ull << (G::W256 & 63)) - 1ull);
    uint32_t mine = 0, mine256 = 0;
    const bool first256 = wave == 0;  // only wave 0 holds samples of the first 256 (uniform)
    for (int b = first256 ? 0 : first_plane; b < nplanes; ++b) {
        uint32_t tot = 0, tot256 = 0;
#pragma unroll
        for (int l = 0; l < G::LV; ++l) {
            const uint64_t m = __ballot((th.cs[l] >> b) & 1u);
            tot += (uint32_t)__popcll(m) << l;
            if (first256) tot256 += (uint32_t)__popcll(m & m256) << l;
        }
        if (lane == b) {
            mine = tot;
            mine256 = tot256;
        }
    }
    if (lane < nplanes) {
        if (lane >= first_plane) atomicAdd(&planeTot[lane], mine);
        if (wave == 0) atomicAdd(&planeTot256[lane], mine256);
    }
}

// The same information for blocks whose prefix sums fit 32 bits (all but loud 24-bit material), without ballots: what
// the scoring needs is A_k = sum_j (u_j >> k) for k = 0..15, and a thread's own A_k follows from its bit-sliced plane
// counts as sum over the slices l of (cs[l] >> k) << l -- ten operations per k.  The sixteen values are summed over the
// wave on the DPP network and lane 63 adds them to the block's (ksum[k]; ksum256[k], k <= 12, for the first 256
// samples: the lanes 0 .. W256-1 of wave 0, an intermediate of the same scan).  About 260 vector instructions per
// wave and candidate against ninety ballot -> scalar round trips (measured: 25 000 -> 10 000 cycles per candidate).
// kmask (wave-uniform): the parameters k whose block sum the scoring can need (static_k_candidates); the others are
// skipped -- except in wave 0, whose first DPP row also feeds the sums over the first 256 samples (k <= 12, all needed by
// the initial-k estimate): there the row sum is still formed and the rest of the reduction left out.
template <class G>
__device__ __forceinline__ void ksums_wave(const Thread<G>& th, uint32_t* ksum, uint32_t* ksum256, int tid, uint32_t kmask = 0xFFFFu) {
    const int lane = tid & 63, wave = tid >> 6;
    static_assert(G::W256 == 16 || G::W256 == 64, "the first 256 samples are one DPP row or the whole wave");
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const bool wanted = ((kmask >> k) & 1u) != 0u;                 // (uniform)
        const bool head256 = wave == 0 && k <= 12;                     // (uniform)
        if (!wanted && !head256) continue;
        uint32_t v = 0;
#pragma unroll
        for (int l = 0; l < G::LV; ++l) v += (th.cs[l] >> k) << l;
        v += dpp_mov<kDppRowShr1, 0xF>(0u, v);
        v += dpp_mov<kDppRowShr2, 0xF>(0u, v);
        v += dpp_mov<kDppRowShr4, 0xF>(0u, v);
        v += dpp_mov<kDppRowShr8, 0xF>(0u, v);
        const uint32_t row = v;  // lane 15: the sum over lanes 0..15
        if (wanted || G::W256 == 64) {
            v += dpp_mov<kDppRowBcast15, 0xA>(0u, v);
            v += dpp_mov<kDppRowBcast31, 0xC>(0u, v);
            if (lane == 63) {
                if (wanted) atomicAdd(&ksum[k], v);
                if (G::W256 == 64 && wave == 0 && k <= 12) atomicAdd(&ksum256[k], v);
            }
        }
        if (G::W256 == 16 && wave == 0 && lane == 15 && k <= 12) atomicAdd(&ksum256[k], row);
    }
}

// The static parameters k that can still be the argmin of A_k + n (1 + k), A_k = sum_j (u_j >> k), given only the block's
// S = sum_j u_j and n (ref block/encoder.cpp:160-180 evaluates all sixteen): floor-of-sum bounds each A_k,
//   (S - n (2^k - 1)) / 2^k <= A_k <= S >> k,
// so a k whose lower cost bound exceeds the smallest upper bound over all k can neither be the minimum nor tie with it.
// Typically four of the sixteen remain (k* - 1 .. k* + 2 around S / n).  Returns the mask of the k to evaluate.
__device__ __forceinline__ uint32_t static_k_candidates(uint64_t S, uint32_t n, int lane) {
    const uint32_t k = (uint32_t)lane & 15u;
    const uint64_t fixed = (uint64_t)n * (1u + k);
    const uint64_t upper = (S >> k) + fixed;
    const uint64_t slack = (uint64_t)n * ((1u << k) - 1u);
    const uint64_t lower = (S > slack ? ((S - slack + ((1u << k) - 1u)) >> k) : 0ull) + fixed;
    const uint64_t umin = wave_last_u64(wave_scan_min_u64(lane < 16 ? upper : ~0ull));
    return (uint32_t)__ballot(lane < 16 && lower <= umin) & 0xFFFFu;
}

// Exclusive scan of an LDS array by one wave (row of 64 at a time, running carry).
__device__ __forceinline__ void wave_exclusive_scan_u32(uint32_t* arr, int len, int lane) {
    uint32_t carry = 0;
    for (int base = 0; base < len; base += 64) {
        const int i = base + lane;
        const uint32_t v = (i < len) ? arr[i] : 0u;
        const uint32_t inc = wave_scan_add_u32(v);
        if (i < len) arr[i] = carry + inc - v;
        carry += wave_last_u32(inc);
    }
}

// ---------------------------------------------------------------------------------------------
// slot geometry
// ---------------------------------------------------------------------------------------------
struct SlotGeom {
    int64_t start;  // first frame (shard-relative)
    uint32_t n;     // frames in the slot
    bool defined;
};

__device__ __forceinline__ uint32_t block_frames(const AnalyzeParams& prm, uint32_t blk) {
    const uint64_t rem = prm.frames - (uint64_t)blk * kMaxBlock;
    return rem < (uint64_t)kMaxBlock ? (uint32_t)rem : (uint32_t)kMaxBlock;
}

__device__ __forceinline__ SlotGeom slot_geom(const AnalyzeParams& prm, uint32_t blk, int slot) {
    SlotGeom g;
    const uint32_t nb = block_frames(prm, blk);
    const int64_t bstart = (int64_t)blk * kMaxBlock;
    const int win = slot >> 2, ch = slot & 3;
    g.defined = true;
    if (prm.channels == 1 && ch != 0) g.defined = false;
    if (win == 0) {
        g.start = bstart;
        g.n = nb;
    } else {
        // probe windows exist only for per-block stereo on blocks above the full-comparison limit
        if (prm.channels != 2 || prm.stereo_mode != 2 || nb <= (uint32_t)kFullCompareLimit) g.defined = false;
        g.n = kProbe;
        g.start = bstart;
        if (g.defined) {
            if (win == 2) g.start = bstart + (nb - kProbe) / 2u;
            if (win == 3) g.start = bstart + nb - kProbe;
        }
    }
    return g;
}

// The stream a global block of the launch set belongs to (StreamDesc, lacx_types.h): the descriptor in the kernel
// arguments when the set is one stream, else a binary search over the table's first blocks (a handful of steps, the same
// for every lane of a workgroup that works on one block).  Returned BY VALUE: the caller's copy lives in (scalar)
// registers; a reference that may point at the kernel arguments or at the table made every later field access a memory
// load (measured: +7 % on the whole-block analysis kernel).
// A descriptor fetched from the table is the same in every lane of the wave (it only depends on the workgroup's block),
// which the compiler cannot see through the search loop: without help it keeps the 26 words in vector registers and the
// analysis kernel spills.  readfirstlane puts them where they belong.
__device__ __forceinline__ StreamDesc wave_uniform(const StreamDesc& d) {
    static_assert(sizeof(StreamDesc) % 4 == 0, "whole words");
    StreamDesc r;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&d);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (size_t i = 0; i < sizeof(StreamDesc) / 4; ++i) dst[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)src[i]);
    return r;
}

// (per lane: kernels whose lanes work on different blocks -- k_stereo, k_levinson, k_offsets)
__device__ __forceinline__ StreamDesc stream_of_block(const BatchRef& br, uint32_t gblk) {
    if (br.table == nullptr) return br.single;
    uint32_t lo = 0, hi = br.nstreams;  // invariant: table[lo].first_block <= gblk < table[hi].first_block
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (br.table[mid].first_block <= gblk) lo = mid; else hi = mid;
    }
    return br.table[lo];
}
// (the whole workgroup works on block gblk)
__device__ __forceinline__ StreamDesc stream_of_block_uniform(const BatchRef& br, uint32_t gblk) {
    if (br.table == nullptr) return br.single;
    return wave_uniform(stream_of_block(br, gblk));
}
// ... by workgroup of the whole-block analysis grid (channels per block: mono and stereo streams share the grid)
__device__ __forceinline__ StreamDesc stream_of_workgroup(const BatchRef& br, uint32_t wg) {
    if (br.table == nullptr) return br.single;
    uint32_t lo = 0, hi = br.nstreams;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (br.table[mid].first_wg <= wg) lo = mid; else hi = mid;
    }
    return wave_uniform(br.table[lo]);
}

// ... by stream index (block * channels + channel over the set; prm.stream_base = the stream's first)
__device__ __forceinline__ StreamDesc stream_of_item(const BatchRef& br, uint32_t item) {
    if (br.table == nullptr) return br.single;
    uint32_t lo = 0, hi = br.nstreams;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (br.table[mid].prm.stream_base <= item) lo = mid; else hi = mid;
    }
    return wave_uniform(br.table[lo]);  // (callers: one item per workgroup)
}

__device__ __forceinline__ SlotSrc slot_src(const AnalyzeParams& prm, const int32_t* L, const int32_t* R, int ch) {
    SlotSrc s;
    s.kind = ch;
    s.a = L;  // planar: left; interleaved: the WAV data chunk
    s.b = R;
    s.layout = prm.layout;
    s.channels = prm.channels;
    return s;
}


// XCD-aware slot mapping.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs, so ids w and w + 8
// share an XCD and its L2.  The `per` workgroups that work on the same block (its channels) all read the same
// PCM, so they get ids 8 apart: the block is fetched from HBM once and the other reads hit that L2.
// Ids beyond the last full group of 8 blocks fall back to the plain (w / per, w % per) mapping.
__device__ __forceinline__ void xcd_slot(uint32_t w, uint32_t per, uint32_t nblocks, uint32_t& blk, uint32_t& which) {
    const uint32_t group = 8u * per, g = w / group, r = w % group;
    if (g < nblocks / 8u) {
        blk = g * 8u + (r & 7u);
        which = r >> 3;
    } else {
        blk = w / per;
        which = w % per;
    }
}

// Hand-off records of the fused emit (producer: k_analyze.hip, consumers: k_emit.hip); see "Fused emit + streaming packer".
constexpr unsigned long long kRecValid = 1ull << 62, kRecMs = 1ull << 61, kRecFlag = 1ull << 60, kRecBytesMask = (1ull << 60) - 1ull;

__device__ __forceinline__ unsigned long long rec_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rec_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// Workgroup barrier that orders LDS accesses only.  __syncthreads() also waits for the wave's global stores
// (s_waitcnt vmcnt(0)); between the output tiles of the emit that would park every wave until its stores have crossed
// PCIe into the pinned host buffer, although nothing on the device reads them.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace lacx
