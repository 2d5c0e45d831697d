// kernels.hip -- CDNA4 (gfx950) kernels of the LAC block-encode analysis path.
//
// Pipeline per shard of 16384-frame blocks (all launches asynchronous on one stream):
//   k_ingest    one workgroup per (block, channel): sample-range validation, the proxy sums of
//               estimate_stereo_mode, exact 13-lag int64 autocorrelation of the whole block and of the
//               3 probe windows                             (ref lac/encoder.cpp:82-102,126-178; lpc.cpp:80-96)
//   k_stereo    sixteen lanes per block: LR/MS estimate -> BlockPlan, need masks (ref lac/encoder.cpp:179-196)
//   k_levinson  one lane per slot: Levinson-Durbin in software x87 extended precision -> Q15 sets
//                                                           (ref lpc.cpp:98-186)
//   k_analyze<4,64>     one wave per probe slot of an "uncertain" block (ref lac/encoder.cpp:341-354)
//   k_decide(1) probes -> LR/MS choice, marks the two whole-block slots still to be analysed
//   k_analyze<16,1024>  one 1024-thread workgroup per needed whole-block slot
//                                                           (ref block/encoder.cpp:313-552)
//                       + the bit emit of its channel block into a staging slot (ref block/encoder.cpp:554-838)
//   k_stream_out        beside it, on its own stream: 32 packer waves move the slots to the (pinned host) payload
//   k_decide(2) small-block full comparison (ref lac/encoder.cpp:336-340), final BlockPlan (only for such a block)
//   k_offsets   one workgroup: block byte sizes -> payload offsets + the container's block table
//   k_pack, k_emit<16,1024>   repair paths: slots the packer did not move / channel blocks the fused emit left out
//                       (k_emit alone is the whole emit with LACX_FUSED_EMIT=0)
//   k_gather    block plans, block table, totals and flags into pinned host memory
// The host emit (emit.cpp, LACX_FLAG_HOST_EMIT) consumes the same ChannelPlan records instead of k_offsets/k_emit.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>

#include "analyze_core.h"
#include "emit_core.h"
#include "kernels.h"

namespace lacx {

// Diagnostic phase stamps (only in builds made with -DLACX_STAMPS; the production kernel has none).
#ifdef LACX_STAMPS
__device__ unsigned long long g_stamp_acc[40];
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
template <class G>
__device__ __forceinline__ void plane_totals_wave(const Thread<G>& th, uint32_t* planeTot, uint32_t* planeTot256,
                                                  int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t any = 0;
#pragma unroll
    for (int l = 0; l < G::LV; ++l) any |= th.cs[l];
    const int nplanes = 32 - __clz((int)wave_or_u32(any));  // uniform, 0..30
    // lanes of wave 0 whose chunk lies inside the first 256 samples
    const uint64_t m256 = (G::W256 >= 64) ? ~0ull : ((1ull << (G::W256 & 63)) - 1ull);
    uint32_t mine = 0, mine256 = 0;
    const bool first256 = wave == 0;  // only wave 0 holds samples of the first 256 (uniform)
    for (int b = 0; b < nplanes; ++b) {
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
        atomicAdd(&planeTot[lane], mine);
        if (wave == 0) atomicAdd(&planeTot256[lane], mine256);
    }
}

// The same information for blocks whose prefix sums fit 32 bits (all but loud 24-bit material), without ballots: what
// the scoring needs is A_k = sum_j (u_j >> k) for k = 0..15, and a thread's own A_k follows from its bit-sliced plane
// counts as sum over the slices l of (cs[l] >> k) << l -- ten operations per k.  The sixteen values are summed over the
// wave on the DPP network and lane 63 adds them to the block's (ksum[k]; ksum256[k], k <= 12, for the first 256
// samples: the lanes 0 .. W256-1 of wave 0, an intermediate of the same scan).  About 260 vector instructions per
// wave and candidate against ninety ballot -> scalar round trips (measured: 25 000 -> 10 000 cycles per candidate).
template <class G>
__device__ __forceinline__ void ksums_wave(const Thread<G>& th, uint32_t* ksum, uint32_t* ksum256, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    static_assert(G::W256 == 16 || G::W256 == 64, "the first 256 samples are one DPP row or the whole wave");
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        uint32_t v = 0;
#pragma unroll
        for (int l = 0; l < G::LV; ++l) v += (th.cs[l] >> k) << l;
        v += dpp_mov<kDppRowShr1, 0xF>(0u, v);
        v += dpp_mov<kDppRowShr2, 0xF>(0u, v);
        v += dpp_mov<kDppRowShr4, 0xF>(0u, v);
        v += dpp_mov<kDppRowShr8, 0xF>(0u, v);
        const uint32_t row = v;  // lane 15: the sum over lanes 0..15
        v += dpp_mov<kDppRowBcast15, 0xA>(0u, v);
        v += dpp_mov<kDppRowBcast31, 0xC>(0u, v);
        if (lane == 63) {
            atomicAdd(&ksum[k], v);
            if (G::W256 == 64 && wave == 0 && k <= 12) atomicAdd(&ksum256[k], v);
        }
        if (G::W256 == 16 && wave == 0 && lane == 15 && k <= 12) atomicAdd(&ksum256[k], row);
    }
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


// ---------------------------------------------------------------------------------------------
// k_ingest: one workgroup per (block, channel in L,R,M,S)
//   * span loads of the channel (M/S derived on the fly), 16 consecutive samples per thread and 4096-sample tile,
//     neighbours' history through LDS, 13-lag exact int64 autocorrelation of the whole block (lpc.cpp:80-96);
//   * the three 256-frame probe windows (lac/encoder.cpp:343-346) as three more small passes;
//   * the channel's three proxy sums of estimate_stereo_mode (lac/encoder.cpp:146-178) and the sample
//     range validation (lac/encoder.cpp:82-102).
// k_stereo: sixteen lanes per block turn the 12 sums into the LR/MS estimate + need masks.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t zz64(int64_t v) {  // ref lac/encoder.cpp:38-41
    return v >= 0 ? ((uint64_t)v << 1) : ((((uint64_t)(-(v + 1))) << 1) | 1u);
}

__device__ __forceinline__ uint64_t approx_rice_bits(uint64_t sum, uint64_t count) {  // ref lac/encoder.cpp:43-57
    if (count == 0) return 0;
    const uint64_t mean = (sum + (count >> 1)) / count;
    // the smallest k <= 31 with 2^k >= mean (the reference counts up from 0)
    uint32_t k = mean <= 1u ? 0u : 64u - (uint32_t)__clzll((long long)(mean - 1u));
    k = k > 31u ? 31u : k;
    return (sum >> k) + count * (uint64_t)(k + 1u);  // saturation is unreachable for <= 2^14 samples of <= 2^27
}

constexpr int kIngestThreads = 256;
constexpr int kIngestTile = 4096;
static_assert(kIngestThreads == kProbe, "one probe sample per thread");

__device__ __forceinline__ bool slot_channel_used(const AnalyzeParams& prm, int ch) {
    if (prm.channels == 1) return ch == 0;
    if (prm.stereo_mode == 0) return ch < 2;
    if (prm.stereo_mode == 1) return ch >= 2;
    return true;
}

// Block-wide sum of 13 per-thread int64 partials into out[13] (global), via wave shuffles + LDS atomics.
__device__ __forceinline__ void reduce13(const int64_t* acc, unsigned long long* s_ac, int64_t* out, int tid) {
    if (tid < 13) s_ac[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        const uint64_t t = wave_sum_u64((uint64_t)acc[k]);
        if ((tid & 63) == 0) atomicAdd(&s_ac[k], (unsigned long long)t);
    }
    __syncthreads();
    if (tid < 13) out[tid] = (int64_t)s_ac[tid];
    __syncthreads();
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

__global__ __launch_bounds__(kIngestThreads, 5) void k_ingest(BatchRef br, unsigned long long* __restrict__ sums,
                                                           uint32_t* __restrict__ badidx,
                                                           int64_t* __restrict__ acorr) {
    // One tile = 4096 samples = 16 consecutive samples per thread, kept in LDS as 4-sample groups in four planes:
    // group g of the tile (samples 4g..4g+3) sits in plane g % 4 at index g / 4 (+1: index 0 of a plane is the
    // group carried over from the previous tile).  Thread t writes its groups 4t..4t+3 -- one 16-byte store per
    // plane, consecutive lanes at consecutive slots -- and, for the 12 samples of history its first group needs,
    // reads groups 4t-3..4t-1 = planes 1..3 at index t-1: every access is conflict-free.
    __shared__ int4 s_plane[4][kIngestThreads + 1];
    __shared__ __align__(16) int32_t s_win[3][12 + kProbe];
    __shared__ unsigned long long s_ac[13];
    __shared__ unsigned long long s_sum[3];
    __shared__ unsigned int s_bad;
    uint32_t blk, chsel;  // blk: global block of the launch set (indexes the workspace)
    xcd_slot(blockIdx.x, 4u, gridDim.x >> 2, blk, chsel);
    const StreamDesc sd = stream_of_block_uniform(br, blk);
    const AnalyzeParams prm = sd.prm;
    const int32_t* __restrict__ L = sd.left;
    const int32_t* __restrict__ R = sd.right;
    const uint32_t lblk = blk - sd.first_block;  // the stream's own block number (geometry, sample addresses)
    const int ch = (int)chsel;
    const bool used = slot_channel_used(prm, ch);
    // forced mid/side still validates the left/right samples (ref lac/encoder.cpp:238-241)
    // ... unless the container cannot hold an out-of-range value: 16-bit containers, packed 24-bit ones at depth 24
    const bool container_bounds = prm.layout == PCM_INTERLEAVED_I16 || (prm.layout == PCM_INTERLEAVED_I24 && prm.bit_depth == 24);
    const bool validate = ch < 2 && ch < prm.channels && prm.bit_depth != 0 && !container_bounds;
    if (!used && !validate) {  // uniform
        if (ch < 2 && threadIdx.x == 0) badidx[blk * 2 + ch] = 0xFFFFFFFFu;
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t nb = block_frames(prm, lblk);
    const int64_t bstart = (int64_t)lblk * kMaxBlock;
    const SlotSrc src = slot_src(prm, L, R, ch);
    const bool est = prm.channels == 2 && prm.stereo_mode == 2;
    const int32_t lo = prm.bit_depth == 16 ? -32768 : -0x800000;
    const int32_t hi = prm.bit_depth == 16 ? 32767 : 0x7FFFFF;
    if (tid < 3) s_sum[tid] = 0;
    if (tid == 0) s_bad = 0xFFFFFFFFu;
    // history before the block start counts as absent (lags start at n = k): the carried groups start as zero
    if (tid < 4) s_plane[tid][0] = make_int4(0, 0, 0, 0);
    __syncthreads();

    constexpr int kPerThread = kIngestTile / kIngestThreads;
    static_assert(kPerThread == 16, "four 4-sample groups per thread and tile");
    int64_t acc[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) acc[k] = 0;
    uint64_t sraw = 0, sdif = 0, sant = 0;
    uint32_t bad = 0xFFFFFFFFu;
    for (uint32_t base = 0; base < nb; base += kIngestTile) {
        // the thread's 16 samples: one span fetch (16-byte loads) in every layout
        const uint32_t first = base + 16u * (uint32_t)tid;
        const int rem = (int)nb - (int)first;
        const int cnt = rem < 0 ? 0 : (rem > 16 ? 16 : rem);
        int32_t w[28];  // w[12 + i] = sample first + i, w[0..11] = the 12 samples before
        load_chunk<16>(src, bstart + first, cnt, bstart + (int64_t)nb - 1, w + 12);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i >= cnt) w[12 + i] = 0;  // past the block end: adds nothing to any lag
            else if (validate && (w[12 + i] < lo || w[12 + i] > hi)) bad = bad < first + (uint32_t)i ? bad : first + (uint32_t)i;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            s_plane[c][tid + 1] = make_int4(w[12 + 4 * c], w[13 + 4 * c], w[14 + 4 * c], w[15 + 4 * c]);
        __syncthreads();
        int4 carry = make_int4(0, 0, 0, 0);
        if (tid >= 1 && tid <= 3) carry = s_plane[tid][kIngestThreads];  // last groups of the tile, for the next one
#pragma unroll
        for (int c = 1; c < 4; ++c) {
            const int4 h = s_plane[c][tid];  // group 4(t-1)+c
            w[4 * (c - 1)] = h.x;
            w[4 * (c - 1) + 1] = h.y;
            w[4 * (c - 1) + 2] = h.z;
            w[4 * (c - 1) + 3] = h.w;
        }
        if (used) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
#pragma unroll
                for (int k = 0; k < 13; ++k) acc[k] += (int64_t)w[12 + i] * (int64_t)w[12 + i - k];
            }
            if (est) {
                // 32-bit zigzags and per-tile 32-bit partial sums: for samples inside the validated range
                // |x| <= 2^24 (mid/side included) the differences fit 26 bits and 16 of them 30; out-of-range
                // input only garbles an estimate of a stream that is rejected anyway.
                auto zz32 = [](int32_t v) { return ((uint32_t)v << 1) ^ (uint32_t)(v >> 31); };
                uint32_t traw = 0, tdif = 0, tant = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (i < cnt) {
                        const int32_t x0 = w[12 + i], prev = w[11 + i];
                        const uint32_t raw = zz32(x0);
                        const bool head = (first + (uint32_t)i) == 0u;
                        traw += raw;
                        tdif += head ? raw : zz32((int32_t)((uint32_t)x0 - (uint32_t)prev));
                        tant += head ? raw : zz32((int32_t)((uint32_t)x0 + (uint32_t)prev));
                    }
                }
                sraw += traw;
                sdif += tdif;
                sant += tant;
            }
        }
        __syncthreads();
        if (tid >= 1 && tid <= 3) s_plane[tid][0] = carry;
        // (the next tile's stores do not touch index 0; its barrier orders this store before thread 0's read)
    }
    if (used) reduce13(acc, s_ac, acorr + ((size_t)blk * kSlotsPerBlock + ch) * 13, tid);

    // probe windows (per-block stereo, blocks above the full-comparison limit only): 256 samples each, lags inside the
    // window only.  All threads stage the three windows; then wave w sums window w on its own -- four samples per lane,
    // one wave reduction per lag, no workgroup-wide reduction and no further barrier.
    if (est && nb > (uint32_t)kFullCompareLimit) {
        if (tid < 36) s_win[tid / 12][tid % 12] = 0;  // the samples before a window count as absent
        for (int w = 1; w <= 3; ++w) {
            const SlotGeom g = slot_geom(prm, lblk, w * 4 + ch);
            s_win[w - 1][12 + tid] = slot_fetch(src, g.start + tid);  // kIngestThreads == kProbe
        }
        __syncthreads();
        const int wv = tid >> 6;
        if (wv < 3) {  // uniform per wave
            int32_t v[16];  // v[12 + i] = sample 4 * lane + i of the window
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int4 q = *reinterpret_cast<const int4*>(&s_win[wv][4 * lane + 4 * c]);
                v[4 * c] = q.x;
                v[4 * c + 1] = q.y;
                v[4 * c + 2] = q.z;
                v[4 * c + 3] = q.w;
            }
            int64_t* out = acorr + ((size_t)blk * kSlotsPerBlock + (wv + 1) * 4 + ch) * 13;
#pragma unroll
            for (int k = 0; k < 13; ++k) {
                int64_t a = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) a += (int64_t)v[12 + i] * (int64_t)v[12 + i - k];
                const uint64_t t = wave_sum_u64((uint64_t)a);
                if (lane == 0) out[k] = (int64_t)t;
            }
        }
    }

    if (est) {
        const uint64_t t0 = wave_sum_u64(sraw), t1 = wave_sum_u64(sdif), t2 = wave_sum_u64(sant);
        if (lane == 0) {
            atomicAdd(&s_sum[0], (unsigned long long)t0);
            atomicAdd(&s_sum[1], (unsigned long long)t1);
            atomicAdd(&s_sum[2], (unsigned long long)t2);
        }
    }
    if (bad != 0xFFFFFFFFu) atomicMin(&s_bad, bad);
    __syncthreads();
    if (tid == 0) {
        if (est) {
            sums[(size_t)blk * 12 + ch] = s_sum[0];
            sums[(size_t)blk * 12 + 4 + ch] = s_sum[1];
            sums[(size_t)blk * 12 + 8 + ch] = s_sum[2];
        }
        if (validate || ch < 2) badidx[blk * 2 + ch] = s_bad;
    }
}

// Sixteen lanes per block, four blocks per wave: lanes 0..11 of a block turn one of its 12 proxy sums into bits (one
// 64-bit division each instead of a chain of twelve), lane 0 of the block decides.
constexpr int kStereoLanes = 16;
__global__ __launch_bounds__(64) void k_stereo(BatchRef br, const unsigned long long* __restrict__ sums,
                                               const uint32_t* __restrict__ badidx, BlockPlan* __restrict__ bplans,
                                               uint32_t* __restrict__ need_probe, uint32_t* __restrict__ need_full) {
    const int tid = threadIdx.x, sub = tid & (kStereoLanes - 1), grp = tid & ~(kStereoLanes - 1);
    const uint32_t blk = blockIdx.x * (64 / kStereoLanes) + (uint32_t)(tid / kStereoLanes);
    const bool live = blk < br.total_blocks;  // every lane stays for the shuffles
    const StreamDesc sd = stream_of_block(br, live ? blk : 0u);
    const AnalyzeParams prm = sd.prm;
    const uint32_t nb = live ? block_frames(prm, blk - sd.first_block) : 0u;
    const bool stereo = prm.channels == 2;
    const bool est = stereo && prm.stereo_mode == 2;
    // estimate_channel_proxy_cost: ref lac/encoder.cpp:114-124 -- sums[blk][kind * 4 + channel], kind = raw, diff, anti
    uint64_t bits = 0;
    if (est && live && sub < 12) bits = approx_rice_bits(sums[(size_t)blk * 12 + sub], nb);
    auto from = [&](int lane_in_group) {
        const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)bits, grp + lane_in_group, 64);
        const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(bits >> 32), grp + lane_in_group, 64);
        return ((uint64_t)hi << 32) | lo;
    };
    uint64_t chbits[4];
    bool active = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint64_t raw = from(c), dif = from(4 + c), ant = from(8 + c);
        uint64_t mn = raw < dif ? raw : dif;
        if (ant < mn) mn = ant;
        chbits[c] = mn;
        active = active || (raw < dif) || (ant < dif);
    }
    if (!live || sub != 0) return;
    BlockPlan bp;
    bp.choose_ms = 0;
    bp.uncertain = 0;
    bp.est_ms = 0;
    // first bad sample in the reference's order: the left channel is validated before the right one
    const uint32_t badl = badidx[blk * 2], badr = stereo ? badidx[blk * 2 + 1] : 0xFFFFFFFFu;
    bp.invalid = (badl != 0xFFFFFFFFu) || (badr != 0xFFFFFFFFu);
    bp.first_bad = (badl != 0xFFFFFFFFu) ? badl : (badr != 0xFFFFFFFFu ? (badr | 0x80000000u) : 0xFFFFFFFFu);
    bp.frames = nb;
    bp.pad = 0;
    uint32_t nprobe = 0, nfull = 0;
    if (!stereo) {
        nfull = 1u;
    } else if (prm.stereo_mode == 0) {
        nfull = 0x3u;
    } else if (prm.stereo_mode == 1) {
        nfull = 0xCu;
        bp.choose_ms = 1;
    } else {
        // the decision: ref lac/encoder.cpp:179-196
        const uint64_t lr = chbits[0] + chbits[1], ms = chbits[2] + chbits[3];
        const uint64_t smaller = lr < ms ? lr : ms;
        const uint64_t diff = lr >= ms ? lr - ms : ms - lr;
        bp.est_ms = ms < lr;
        bp.choose_ms = bp.est_ms;
        bp.uncertain = smaller == 0 || diff == 0 || active || diff <= smaller / 100u;
        if (!bp.uncertain) {
            nfull = bp.est_ms ? 0xCu : 0x3u;
        } else if (nb <= (uint32_t)kFullCompareLimit) {
            nfull = 0xFu;  // encode both, compare sizes (k_decide phase 2)
        } else {
            nprobe = 0xFFF0u;  // 12 probe slots; the whole-block pair is picked by k_decide phase 1
        }
    }
    bplans[blk] = bp;
    need_probe[blk] = nprobe;
    need_full[blk] = nfull;
}

// ---------------------------------------------------------------------------------------------
// k_levinson: one lane per slot that needs it
// ---------------------------------------------------------------------------------------------
constexpr int kLevThreads = 256;
struct LevMem {  // work arrays R, a, prevA of every thread, one column per thread (120 KiB)
    uint64_t m[3][13][kLevThreads];
    uint32_t es[3][13][kLevThreads];  // sign << 31 | (exponent + 2^21)
};

// The recursion is a serial chain of ~370 software-float operations per slot, so the kernel's duration is one
// thread's latency whatever the grid looks like.  What the grid decides is how many CUs it takes away from the
// other pipeline chunks' kernels meanwhile: slots are numbered slot-major (waves made of probe slots of certain
// blocks leave at once) and packed 256 to a workgroup.
__global__ __launch_bounds__(kLevThreads) void k_levinson(BatchRef br, const int64_t* __restrict__ acorr,
                                                          const uint32_t* __restrict__ need_probe,
                                                          LpcSet* __restrict__ lpcs) {
    extern __shared__ __align__(16) unsigned char lev_raw[];
    LevMem& lm = *reinterpret_cast<LevMem*>(lev_raw);
    const uint32_t id = blockIdx.x * kLevThreads + threadIdx.x;
    const uint32_t nblk = br.total_blocks;
    const int slot = (int)(id / nblk);
    const uint32_t blk = id % nblk;
    if (slot >= kSlotsPerBlock) return;
    const StreamDesc sd = stream_of_block(br, blk);
    const AnalyzeParams prm = sd.prm;
    const SlotGeom g = slot_geom(prm, blk - sd.first_block, slot);
    if (!g.defined) return;
    if (prm.channels == 2 && slot < 4) {
        if (prm.stereo_mode == 0 && (slot & 3) >= 2) return;
        if (prm.stereo_mode == 1 && (slot & 3) < 2) return;
    }
    if (slot >= 4 && !((need_probe[blk] >> slot) & 1u)) return;  // probe windows of blocks that are not probed
    const uint32_t gid = blk * kSlotsPerBlock + (uint32_t)slot;
    const int mvo = (g.n > 1) ? (int)((g.n - 1 < 32u) ? g.n - 1 : 32u) : 0;
    struct LdsArray {
        uint64_t (*m)[kLevThreads];
        uint32_t (*es)[kLevThreads];
        int lane;
        __device__ xf80 get(int i) const {
            const uint32_t w = es[i][lane];
            return xf80{m[i][lane], (int32_t)(w & 0x7FFFFFFFu) - (1 << 21), w >> 31};
        }
        __device__ void set(int i, xf80 x) {
            m[i][lane] = x.m;
            es[i][lane] = (x.s << 31) | ((uint32_t)(x.e + (1 << 21)) & 0x7FFFFFFFu);
        }
    };
    const int lane = (int)threadIdx.x;
    LdsArray Rv{lm.m[0], lm.es[0], lane}, av{lm.m[1], lm.es[1], lane}, pv{lm.m[2], lm.es[2], lane};
    const int64_t* r = acorr + (size_t)gid * 13;
    LpcSet* out = &lpcs[gid];
    levinson_candidates_t([r](int i) { return r[i]; }, mvo, Rv, av, pv,
                          [out](int ci, int j, int16_t v) { out->coef[ci][j] = v; },
                          [out](int ci, uint8_t v) { out->used[ci] = v; });
    out->pad = 0;
}

// suffix-min scan of tabNX (first non-zero index per chunk) -> exclusive: min over later chunks
template <class G, class M>
__device__ __forceinline__ int32_t scan_nx_part1(M& sh, int tid, int32_t* wtot) {
    const int lane = tid & 63, wave = tid >> 6;
    int32_t v = sh.tabNX[tid];
    int32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t o = __shfl_down(inc, d, 64);
        if (lane + d < 64) inc = o < inc ? o : inc;
    }
    if (lane == 0) wtot[wave] = inc;
    return inc;
}

template <class G, class M>
__device__ __forceinline__ void scan_nx_part2(M& sh, int tid, int32_t inc, const int32_t* wtot, int32_t n) {
    constexpr int NW = G::T / 64;
    const int lane = tid & 63, wave = tid >> 6;
    int32_t later = n;
    for (int w = wave + 1; w < NW; ++w) later = wtot[w] < later ? wtot[w] : later;
    int32_t next = __shfl_down(inc, 1, 64);
    if (lane == 63) next = n;
    sh.tabNX[tid] = next < later ? next : later;
}


// Workgroup barrier that orders LDS accesses only.  __syncthreads() also waits for the wave's global stores
// (s_waitcnt vmcnt(0)); between the output tiles of the emit that would park every wave until its stores have crossed
// PCIe into the pinned host buffer, although nothing on the device reads them.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Emit of one channel block from the residual in sh.u (plain zigzag values, block scans of tabP / tabNZ / tabNX done,
// plan fields loaded): walk 1 (Rice parameter per sample + token bits), bit offsets, walk 2 into 48 KiB LDS tiles,
// copy-out.  `resolve` is called once by all threads (it may contain barriers) before the first byte leaves the
// workgroup and yields the address the channel block's bitstream goes to; false from it abandons the emit.  Returns
// true when the whole bitstream was written.  Shared by k_emit and the emit fused into the analysis kernel.
template <class G, class M, class Resolve>
__device__ __forceinline__ bool emit_body(M& sh, Thread<G>& th, uint32_t n, uint8_t* __restrict__ out,
                                          uint32_t* __restrict__ err_flag, Resolve&& resolve, const int tid,
                                          const bool ablate_stores, const uint32_t slot_bytes STAMP_PARAMS) {
    (void)out;
    const bool narrow = sh.tabP[G::T] < (1ull << 31);
    const bool adaptive0 = sh.p == 0 && (sh.part_mode_k[0] >> 5) != 3;  // stateful Rice::adapt_k walk
    if (adaptive0) {
        if (narrow) {
            phase_a<G, true>(th, sh);
        } else {
            phase_a<G, false>(th, sh);
        }
        __syncthreads();
    }
    STAMP(25);
    auto orw = [](uint32_t* w, uint32_t v) { atomicOr(w, v); };
    auto stw = [](uint32_t* w, uint32_t v) { *w = v; };
    // walk 1: Rice parameter per sample + token bits of the chunk
    const unsigned long long mybits = narrow ? emit_walk<G, true>(th, sh, nullptr, 0, orw, stw)
                                             : emit_walk<G, false>(th, sh, nullptr, 0, orw, stw);
    STAMP(26);
    __syncthreads();  // every thread is done with the sample prefix sums: tabP becomes the bit-offset table
    sh.tabP[tid] = mybits;
    {
        // sum scan of the bit counts (tabP only)
        const int lane = tid & 63, wave = tid >> 6;
        const unsigned long long inc = wave_scan_add_u64(mybits);
        if (lane == 63) sh.wtotP[wave] = inc;
        __syncthreads();
        unsigned long long base = 0;
        for (int w = 0; w < wave; ++w) base += sh.wtotP[w];
        sh.tabP[tid] = base + inc - mybits;
        if (tid == G::T - 1) sh.tabP[G::T] = base + inc;
        __syncthreads();
    }
    const unsigned long long total_bits = sh.tabP[G::T] + sh.header_bits;
    const unsigned long long nbytes = (total_bits + 7u) >> 3;
    if (tid == 0 && (nbytes != sh.payload_bytes || sh.err)) atomicOr(err_flag, 1u);
    if (nbytes != sh.payload_bytes || sh.err) return false;  // uniform: never write outside the planned byte range
    const unsigned long long mypos = sh.tabP[tid] + sh.header_bits;
    STAMP(27);

    // walk 2: tokens into 48 KiB LDS tiles, copied out tile by tile
    uint8_t* base = nullptr;
    for (unsigned long long bit0 = 0; bit0 < nbytes * 8u; bit0 += (unsigned long long)kEmitTileWords * 32u) {
        if (bit0 != 0) lds_barrier();  // every thread has copied its part of the previous tile out of LDS
        {
            // only the words this tile's bytes occupy (+ what the 16-byte copy-out may read past them)
            const unsigned long long left_bytes = nbytes - (bit0 >> 3);
            const int zw = left_bytes >= (unsigned long long)kEmitTileWords * 4u ? kEmitTileWords
                                                                                 : (int)((((uint32_t)left_bytes + 15u) >> 4) * 4u + 8u);
            const int zero_words = zw < kEmitTileWords ? zw : kEmitTileWords;
            for (int i = tid; i < zero_words; i += G::T) sh.xp.o.obits[i] = 0;
        }
        lds_barrier();
        STAMP(28);
        BitTile tile{sh.xp.o.obits, bit0, (uint32_t)kEmitTileWords};
        if (bit0 == 0) emit_header(th, sh, &tile, orw);
        const unsigned long long tile_end = bit0 + (unsigned long long)kEmitTileWords * 32u;
        if (mypos < tile_end && mypos + mybits > bit0) {
            if (narrow) {
                emit_walk<G, true>(th, sh, &tile, mypos, orw, stw);
            } else {
                emit_walk<G, false>(th, sh, &tile, mypos, orw, stw);
            }
        }
        STAMP(29);
        lds_barrier();
        STAMP(30);
        if (bit0 == 0 && !resolve(&base)) return false;  // uniform
        const unsigned long long byte0 = bit0 >> 3;
        const unsigned long long left = nbytes - byte0;
        const uint32_t count = left < (unsigned long long)kEmitTileWords * 4u ? (uint32_t)left : (uint32_t)kEmitTileWords * 4u;
        // Copy-out in 16-byte stores on 16-byte boundaries of the destination (which may be pinned host
        // memory behind PCIe: whole, aligned segments matter there); the unaligned head and tail go bytewise.
        uint8_t* dst = base + byte0;
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u);
        const uint32_t head = mis ? (16u - mis < count ? 16u - mis : count) : 0u;
        const uint32_t nvec = (count - head) >> 4;
        const uint32_t* tw = sh.xp.o.obits;
        auto tile_byte = [&](uint32_t i) { return (uint8_t)(tw[i >> 2] >> (24u - 8u * (i & 3u))); };
        if (ablate_stores) continue;  // timing ablation only (LACX_DEBUG_SKIP bit 11)
        if (slot_bytes) {
            // Staging slot (16-byte aligned, padded): whole 16-byte vectors only, stored write-through (sc1) so that
            // the hand-off to the streaming packer needs no release fence (cdna_hip_programming.md, Guideline 16, R1).
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)slot_bytes, 0x00020000);
            const uint32_t nv = (count + 15u) >> 4;
            for (uint32_t v = tid; v < nv; v += G::T) {
                u32x4 o;
                o.x = __builtin_bswap32(tw[4u * v]);
                o.y = __builtin_bswap32(tw[4u * v + 1u]);
                o.z = __builtin_bswap32(tw[4u * v + 2u]);
                o.w = __builtin_bswap32(tw[4u * v + 3u]);
                __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, (int)((uint32_t)byte0 + 16u * v), 0, 16 /* sc1 */);
            }
            continue;
        }
        if ((uint32_t)tid < head) dst[tid] = tile_byte((uint32_t)tid);
        {
            const uint32_t r = head & 3u, j0 = head >> 2;
            uint4* vdst = reinterpret_cast<uint4*>(dst + head);
            for (uint32_t v = tid; v < nvec; v += G::T) {
                const uint32_t j = j0 + 4u * v;
                uint32_t w[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const uint32_t jj = j + q < (uint32_t)kEmitTileWords ? j + q : (uint32_t)kEmitTileWords - 1u;
                    w[q] = __builtin_bswap32(tw[jj]);  // bytes of the stream in memory order
                }
                uint4 o;
                o.x = __builtin_amdgcn_alignbyte(w[1], w[0], r);
                o.y = __builtin_amdgcn_alignbyte(w[2], w[1], r);
                o.z = __builtin_amdgcn_alignbyte(w[3], w[2], r);
                o.w = __builtin_amdgcn_alignbyte(w[4], w[3], r);
                vdst[v] = o;
            }
        }
        {
            const uint32_t t0 = head + (nvec << 4);
            if (t0 + (uint32_t)tid < count) dst[t0 + tid] = tile_byte(t0 + (uint32_t)tid);
        }
        // no barrier and no wait for the stores here: the workgroup may retire while they are still on their way
        STAMP(31);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// Fused emit + streaming packer.
// The whole-block analysis kernel emits a channel block's bitstream right after its plan is final, while the winner's
// residual (plain, in sh.u) and its block scans are still in LDS -- no re-staging of the PCM, no second residual pass.
// Where the bytes belong in the shard payload depends on the sizes of all earlier channel blocks, so the workgroup
// writes them to the channel block's staging slot in device memory (fixed stride, 16-byte aligned) and retires: no
// analysis workgroup ever waits for another one.  A small companion kernel, k_stream_out, runs beside the analysis
// on its own stream: it walks the stream indices in order, waits for each slot to be published, keeps the running
// byte offset and copies slot after slot to its place in the payload (pinned host memory: the bytes cross PCIe while
// later blocks are still being analysed, nothing is left to copy when the analysis ends).
// Hand-off per stream index i (= block * channels + channel), two 8-byte words, each written by ONE agent-scope store:
//   size_rec[i]  = 1 << 62 | ms << 61 | flag byte in front << 60 | bytes
//                                                 as soon as the plan is final (the data is the flag: R2 granule of
//                                                 MI355X_MICROARCH.md, no fence needed)
//   ready_rec[i] = 1  the bitstream is in slot i: the slot is written with write-through (sc1) stores and announced
//                     behind every storing wave's s_waitcnt vmcnt(0) and the workgroup barrier (cdna_hip_programming.md,
//                     Guideline 16, R1); the consumer polls relaxed, then fences with an agent-scope acquire before
//                     it reads the slot;
//                  2  no bitstream will come from the analysis kernel (left to k_emit).
// Every wait of the packer is bounded; when it gives up, or for anything it did not move, k_pack / k_emit finish the
// job after the analysis (they always run), so no dispatch order or co-residency is assumed for correctness.
// ---------------------------------------------------------------------------------------------
constexpr unsigned long long kRecValid = 1ull << 62, kRecMs = 1ull << 61, kRecFlag = 1ull << 60, kRecBytesMask = (1ull << 60) - 1ull;

__device__ __forceinline__ unsigned long long rec_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rec_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// idx: stream index of this channel block; flag_byte: the block's LR/MS flag byte precedes this channel block.
template <class G>
__device__ __forceinline__ void fused_emit(Smem<G>& sh, Thread<G>& th, const AnalyzeParams& prm, const FuseArgs& fa,
                                           const long long idx, const bool flag_byte, const uint32_t flag_value,
                                           const int tid STAMP_PARAMS) {
    const uint32_t n = th.n;
    // Optimisation barrier on the thread's coordinates: without it the compiler computes the LDS addresses of the emit
    // phases at kernel entry and keeps them alive (spilled to scratch) through the whole analysis.
    asm volatile("" : "+v"(th.tid), "+v"(th.a));
    const unsigned long long my_size = (unsigned long long)sh.plan.payload_bytes + (flag_byte ? 1u : 0u);
    // the size is final: publish it at once (the packer can account for this block while it is being emitted)
    if (tid == 0) rec_store(&fa.size_rec[idx], kRecValid | (flag_value ? kRecMs : 0ull) | (flag_byte ? kRecFlag : 0ull) | my_size);
    // a bitstream longer than the slot (never seen: it would take > 3 resp. 5 bytes per sample) is left to k_emit;
    // test hook (LACX_DEBUG_SKIP bit 10): so is every fifth channel block
    const bool skip = (unsigned long long)sh.plan.payload_bytes + 16u > fa.slot_stride ||
                      ((prm.debug_skip & 1024u) && (idx % 5 == 3));
    bool done = false;
    if (!skip) {  // uniform
        emit_load_plan(sh, sh.plan, tid, G::T);
        __syncthreads();
        // the "first non-zero sample after me" table is only read by zero-run partitions (mode 1)
        if (sh.plan_any_zr) {  // uniform
            emit_first_nonzero(th, sh);
            const int32_t nxinc = scan_nx_part1<G>(sh, tid, sh.wx);
            __syncthreads();
            scan_nx_part2<G>(sh, tid, nxinc, sh.wx, (int32_t)n);
            __syncthreads();
        }
        uint8_t* slot = fa.slots + (unsigned long long)idx * fa.slot_stride;
        STAMP(24);
        done = emit_body<G>(sh, th, n, slot, fa.err_flag, [slot](uint8_t** o) { *o = slot; return true; }, tid,
                            (prm.debug_skip & 2048u) != 0u, (uint32_t)fa.slot_stride STAMP_ARGS);
    }
    // publish: the slot was written with write-through (sc1) stores; every storing wave drains them, the workgroup
    // meets, then one lane announces the slot (no release fence needed for sc1 payload: Guideline 16, R1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (done) fa.emitted[idx] = 2u;  // for the kernels that run after this one (k_pack)
        // test hook (bit 13): the slot is filled but never announced, so the packer gives up and k_pack takes over
        if (!(prm.debug_skip & 8192u)) rec_store(&fa.ready_rec[idx], done ? 1ull : 2ull);
    }
}

// Candidate scoring (ref block/encoder.cpp:337-359) by the 64 lanes of one wave; same result as score_candidate() of
// analyze_core.h, which one thread computes in a serial chain of ~150 dependent 64-bit operations while fifteen
// waves wait for it at the next barrier.  Static Rice cost at k: sum_j (u_j >> k) = T_k >> k with
// T_k = sum_{b >= k} C_b << b, a suffix sum over the bit-plane counts: lane l takes plane 29 - l, one prefix scan gives
// every T_k, lanes 14..29 hold k = 15..0; the (cost, k) minimum with ties to the lower k is a minimum of cost * 16 + k.
template <class G>
__device__ __forceinline__ void score_candidate_wave(Smem<G>& sh, int cand, uint32_t n, int zero_run, uint32_t k0,
                                                     const uint32_t* planeTot, const unsigned long long* acc, int lane,
                                                     bool ksums) {
    uint64_t key = ~0ull;
    if (ksums) {  // planeTot[k] = sum_j (u_j >> k) already (ksums_wave)
        if (lane < 16) key = (((uint64_t)planeTot[lane] + (uint64_t)n * (uint64_t)(1 + lane)) << 4) | (uint64_t)lane;
    } else {
        const int b = 29 - lane;
        const uint64_t w = (lane < 30) ? ((uint64_t)planeTot[b] << b) : 0ull;
        const uint64_t tk = wave_scan_add_u64(w);  // lane l: T_(29-l)
        if (lane >= 14 && lane < 30) key = (((tk >> b) + (uint64_t)n * (uint64_t)(1 + b)) << 4) | (uint64_t)b;  // cost < 2^45
    }
    const uint64_t best_key = wave_last_u64(wave_scan_min_u64(key));
    if (lane == 0) {
        const uint64_t sbits = best_key >> 4;
        const uint32_t sk = (uint32_t)(best_key & 15u);
        const uint64_t rice = acc[0], bin = acc[1];
        const uint32_t hasrun = acc[3] != 0;
        const uint64_t zr = (zero_run && hasrun) ? acc[2] : rice;
        const uint64_t a = rice < sbits ? rice : sbits;
        const uint64_t c = zr < bin ? zr : bin;
        const uint64_t best = a < c ? a : c;
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
}

// estimate_initial_k (ref block/encoder.cpp:121-158) from the plane counts of the first min(256, n) samples, by the 64
// lanes of one wave (same suffix-sum formulation as score_candidate_wave; k = 0..12, ties to the lower k).
__device__ __forceinline__ uint32_t initial_k_wave(const uint32_t* planes256, uint32_t n, int lane, bool ksums) {
    const uint32_t m = n < 256u ? n : 256u;
    uint64_t key = ~0ull;
    if (ksums) {  // planes256[k] = sum over the first min(256, n) samples of u >> k (ksums_wave)
        if (lane <= 12) key = (((uint64_t)planes256[lane] + (uint64_t)m * (uint64_t)(1 + lane)) << 4) | (uint64_t)lane;
    } else {
        const int b = 29 - lane;
        const uint64_t w = (lane < 30) ? ((uint64_t)planes256[b] << b) : 0ull;
        const uint64_t tk = wave_scan_add_u64(w);  // lane l: T_(29-l)
        if (lane >= 17 && lane < 30) key = (((tk >> b) + (uint64_t)m * (uint64_t)(1 + b)) << 4) | (uint64_t)b;
    }
    return (uint32_t)(wave_last_u64(wave_scan_min_u64(key)) & 15u);
}

// ---------------------------------------------------------------------------------------------
// k_analyze
// ---------------------------------------------------------------------------------------------
// The analysis of one slot (everything after the slot has been picked).
template <class G>
__device__ __forceinline__ void analyze_slot(unsigned char* smem_raw, const AnalyzeParams& prm, uint32_t n_in,
                                             const SlotSrc& src, int64_t start, const LpcSet* __restrict__ lpc_slot,
                                             ChannelPlan* __restrict__ plan_out, const int tid, const FuseArgs& fuse,
                                             const long long fuse_idx, const bool fuse_flag_byte,
                                             const uint32_t fuse_flag_value) {
    Smem<G>& sh = *reinterpret_cast<Smem<G>*>(smem_raw);
    const uint32_t n = n_in;
#ifdef LACX_STAMPS
    unsigned long long stamp_acc[40];
    for (int k = 0; k < 40; ++k) stamp_acc[k] = 0;
    unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long stamp_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    Thread<G> th;
    thread_init(th, n, tid);
    stage_samples(th, sh, src, start);
    for (int i = tid; i < (int)(sizeof(LpcSet) / 2); i += G::T)
        reinterpret_cast<uint16_t*>(&sh.lpc)[i] = reinterpret_cast<const uint16_t*>(lpc_slot)[i];
    if (tid < 32) {
        sh.planeTot[0][tid] = sh.planeTot[1][tid] = 0;
        sh.planeTot256[0][tid] = sh.planeTot256[1][tid] = 0;
    }
    if (tid < 4) sh.acc[0][tid] = sh.acc[1][tid] = 0;
    if (tid < 33) (&sh.lbacc[0][0])[tid] = 0;
    if (tid < 2) sh.has4[tid] = 0;
    if (tid == 0) sh.best_cand = -1;
    __syncthreads();
    STAMP(0);

    // ---- pass 1: the pruning bound of every candidate ------------------------------------------------------------
    // One walk over the chunk for all eleven candidates (pass1_bounds, analyze_core.h): window loaded once, nothing
    // stored, no barrier between candidates (ref block/encoder.cpp:362-407 walks them one by one).  Per candidate the
    // thread holds the sum of its leading-bit counts and a 2-bit code per position (zero / four / other) that is counted
    // once per chunk.
    {
        // Optimisation barrier on the chunk origin: without it the compiler hoists a dozen loop-invariant LDS
        // addresses and masks derived from it and, at the 128-VGPR budget, spills them to scratch.
        asm volatile("" : "+v"(th.a));
        BoundPartials bp[11];
        const bool lpc_off = (prm.debug_skip & 16u) != 0u;
        if (n == (uint32_t)G::MAXN) pass1_bounds<G, true>(th, sh, lpc_off, bp);  // uniform
        else pass1_bounds<G, false>(th, sh, lpc_off, bp);
        // bit_width(u | 1) + 1 = 33 - clz(u | 1) = 34 - lead_m per position.  A zero counts 2 that way and is worth 1 (the
        // wave's zero count takes the difference out); a position beyond the slot (r = 0) counts 2, is among those zeros
        // and is worth nothing (the wave's `beyond` takes the rest out).
        // Two candidates share a register for the wave sums (each sum stays below 2^16: 64 lanes x 34 x CH).
        const uint32_t per_thread = 34u * (uint32_t)G::CH;
        const int32_t left = (int32_t)n - (int32_t)((tid >> 6) * 64 * G::CH);  // samples of the slot from this wave's first one on
        const uint32_t valid = left <= 0 ? 0u : (left >= 64 * G::CH ? (uint32_t)(64 * G::CH) : (uint32_t)left);
        const uint32_t beyond = (uint32_t)(64 * G::CH) - valid;
#pragma unroll
        for (int c = 0; c < 12; c += 2) {
            const uint32_t lo = per_thread - bp[c].msum, hi = c + 1 < 11 ? per_thread - bp[c + 1 < 11 ? c + 1 : c].msum : 0u;
            const uint32_t g2 = wave_sum_u32(lo | (hi << 16));
            const uint32_t cnt0 = wave_sum_u32(bound_counts<G::CH>(bp[c]));
            const uint32_t cnt1 = c + 1 < 11 ? wave_sum_u32(bound_counts<G::CH>(bp[c + 1 < 11 ? c + 1 : c])) : 0u;
            if ((tid & 63) == 0) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int cc = c + h;
                    if (cc < 11) {
                        const uint32_t g = h ? (g2 >> 16) : (g2 & 0xFFFFu);
                        const uint32_t cnt = h ? cnt1 : cnt0;
                        const uint32_t nz = cnt & 0x7FFu, n4 = (cnt >> 11) & 0x7FFu, ends = cnt >> 22;
                        atomicAdd(&sh.lbacc[cc][0], g - nz - beyond);
                        atomicAdd(&sh.lbacc[cc][1], (nz - beyond) + (n4 << 16));
                        atomicAdd(&sh.lbacc[cc][2], ends);
                    }
                }
            }
        }
    }
    STAMP(2);
    __syncthreads();
    if (tid <= 10) {
        // one lane per candidate: its bound as a sortable key (bound * 16 + index), all ones when it is not available
        const bool avail = !((tid >= 6 && (sh.lpc.used[tid >= 6 ? tid - 6 : 0] == 0 || (prm.debug_skip & 16u))) ||
                             (tid >= 1 && (prm.debug_skip & 64u)));
        sh.cand_key[tid] = avail ? ((candidate_lower_bound(sh.lbacc[tid][0], sh.lbacc[tid][1], sh.lbacc[tid][2], n, prm.zero_run) << 4) | (uint64_t)tid)
                                 : ~0ull;
    }
    STAMP(5);

    // ---- pass 2: exact costs, most promising candidate first ----------------------------------------------------------
    // The reference keeps the first candidate with the strictly smallest cost = the minimum of (cost, index).  Candidates
    // are evaluated in ascending (bound, index) order; one that cannot beat the best (cost, index) so far ends the search,
    // because every remaining one has a bound at least as large (and, at an equal bound, a larger index).  A dismissed
    // candidate costs nothing here: no residual, no barrier.
    int pending = -1;        // candidate whose totals still have to be scored
    uint32_t pending_k0 = 0;
    bool pending_ksums = false;  // its totals are k-sums (32-bit blocks), not plane counts
    uint32_t tried = 0;      // candidates already evaluated (uniform)
    int parity = 0;
    for (;;) {
        if (tid < 64) {  // wave 0
            if (pending >= 0) {
                // previous candidate's totals sit in the other buffers: score it, then clear them
                score_candidate_wave(sh, pending, n, prm.zero_run, pending_k0, sh.planeTot[parity ^ 1], sh.acc[parity ^ 1], tid, pending_ksums);
                if (tid < 32) sh.planeTot[parity ^ 1][tid] = sh.planeTot256[parity ^ 1][tid] = 0;
                if (tid < 4) sh.acc[parity ^ 1][tid] = 0;
            }
            // next: the untried candidate with the smallest (bound, index), unless it cannot win any more
            const uint64_t key = (tid <= 10 && !((tried >> tid) & 1u)) ? sh.cand_key[tid] : ~0ull;
            const uint64_t best_key = wave_last_u64(wave_scan_min_u64(key));
            if (tid == 0) {
                sh.has4[parity] = 0;
                int next = best_key == ~0ull ? -1 : (int)(best_key & 15u);
                if (next >= 0 && !(prm.debug_skip & 128u) && candidate_pruned(best_key >> 4, next, sh.best_bits, sh.best_cand)) next = -1;
                sh.next_cand = next;
            }
        }
        STAMP(1);
        __syncthreads();  // Bsel: the previous candidate is scored, the next one chosen
        STAMP(4);
        const int cand = sh.next_cand;
        if (cand < 0) break;  // uniform
        tried |= 1u << cand;
        asm volatile("" : "+v"(th.a));  // (see pass 1)
        uint32_t* pt = sh.planeTot[parity];
        uint32_t* pt256 = sh.planeTot256[parity];
        unsigned long long* acc = sh.acc[parity];
        {
            uint32_t ures[G::CH];
            phase_r_residual(th, sh, cand, ures);
            phase_r_store(th, sh, ures);
        }
        ScanRegs<G> sr;
        scan_pz_part1(sh, tid, sr);
        STAMP(3);
        __syncthreads();  // B1b: the wave totals of the scan
        const uint64_t total_u = scan_pz_part2(sh, tid, sr);
        const bool narrow = total_u < (1ull << 31);  // all prefix sums fit 32 bits (uniform)
        const bool ksums = narrow && !(prm.debug_skip & 262144u);
        if (!(prm.debug_skip & 1u)) {
            if (ksums) ksums_wave(th, pt, pt256, tid); else plane_totals_wave(th, pt, pt256, tid);
        }
        // the first 256 samples all belong to wave 0: its own totals are complete once its atomics are (same wave,
        // program order), so it can derive the initial k at once; every other thread reads it after B3
        if (tid < 64) {
            const uint32_t k0w = initial_k_wave(pt256, n, tid, ksums);
            if (tid == 0) sh.cur_k0 = k0w;
        }
        STAMP(6);
        if (prm.debug_skip & 2u) {
            sh.tabF[tid] = 0;
            th.has4 = 1u;
        } else if (narrow) {
            phase_a<G, true>(th, sh);
        } else {
            phase_a<G, false>(th, sh);
        }
        if (__ballot(th.has4 != 0u) != 0ull && (tid & 63) == 0) sh.has4[parity] = 1u;  // read after B3
        STAMP(8);
        __syncthreads();  // B3: every chunk's flag counts are in tabF (phase B sums the six before its own), prefixes in tabP
        STAMP(10);
        const uint32_t k0 = sh.cur_k0;
        if (prm.debug_skip & 4u) {
            th.crice = th.cbin = th.czr = 1;
            th.chasrun = 0;
        } else {
            // the zero-run cost only matters when the residual has a run of >= 4 zeros somewhere
            const bool zr = prm.zero_run && sh.has4[parity] != 0u;
            phase_b_dispatch<G>(th, sh, k0, narrow, zr, n == (uint32_t)G::MAXN);
        }
        STAMP(12);
        {
            const bool active = (uint32_t)th.a < n;
            const uint64_t r0 = wave_sum_u64(active ? th.crice : 0ull);
            const uint64_t r1 = wave_sum_u64(active ? th.cbin : 0ull);
            const uint64_t r2 = wave_sum_u64(active ? th.czr : 0ull);
            const uint32_t r3 = wave_or_u32(active ? th.chasrun : 0u);
            if ((tid & 63) == 0) {
                atomicAdd(&acc[0], (unsigned long long)r0);
                atomicAdd(&acc[1], (unsigned long long)r1);
                atomicAdd(&acc[2], (unsigned long long)r2);
                atomicAdd(&acc[3], (unsigned long long)r3);
            }
        }
        STAMP(13);
        __syncthreads();  // B5
        STAMP(14);
        pending = cand;
        pending_k0 = k0;
        pending_ksums = ksums;
        parity ^= 1;
    }

    STAMP(15);
    // ---- partition search on the winning residual -------------------------------------------
    const int best = sh.best_cand;
    int max_p = 0;
    if (prm.partitioning && n >= (uint32_t)kMinPartition) max_p = max_partition_order(n);
    const int nseg = max_p > 0 ? ((2 << max_p) - 2) : 0;
    PartMem<G>& pm = sh.xp.part;
    auto clear_partition_scratch = [&]() {  // aliases the staged samples: only once every thread is done with them
        for (int i = tid; i < nseg; i += G::T) {
            pm.segacc[i][0] = pm.segacc[i][1] = pm.segacc[i][2] = 0;
            pm.segrun[i] = 0;
        }
        if (tid <= G::MAXP) pm.pbits[tid] = 0;
    };
    if (best == pending && !(prm.debug_skip & 16384u)) {
        // The winner is the candidate evaluated last (usually the only one): its residual is still in sh.u, with the
        // micro-window flags of phase A in bits 30/31, its prefix sums in tabP / tabNZ, its plane counts in th.cs.
        // Strip the flags; nobody reads the staged samples any more (the last barrier of the search is behind us).
#pragma unroll
        for (int i = 0; i < G::CH; ++i) sh.u[i * G::T + tid] &= 0x3FFFFFFFu;
        clear_partition_scratch();
        __syncthreads();
    } else {
        phase_r(th, sh, best);  // last reader of the staged samples; leaves the plain residual in sh.u
        ScanRegs<G> sr;
        scan_pz_part1(sh, tid, sr);
        __syncthreads();
        clear_partition_scratch();
        scan_pz_part2(sh, tid, sr);
        __syncthreads();
    }
    const bool pnarrow = sh.tabP[G::T] < (1ull << 31);
    STAMP(16);
    if (max_p > 0) {
        {
            // the TPG neighbouring lanes of a 64-sample group own its table entry: sum them on the DPP network, one plain
            // store by the group's last lane (no atomics; the entries need no clearing)
            constexpr int kLog = G::TPG == 4 ? 2 : 4;
            static_assert(G::TPG == 4 || G::TPG == 16, "lanes per 64-sample group");
            uint32_t words[15];
            packed_planes(th, words);
            const bool last = (tid & (G::TPG - 1)) == G::TPG - 1;
#pragma unroll
            for (int w = 0; w < 15; ++w) {
                const uint32_t v = seg_sum_u32<kLog>(words[w]);
                if (last) pm.grp[w][tid / G::TPG] = v;
            }
            if (tid < 15) pm.grp[tid][G::NG] = 0;  // the slot past the last group (the scan's total)
        }
        __syncthreads();
        {
            constexpr int NW = G::T / 64;
            const int wave = tid >> 6, lane = tid & 63;
            for (int w = wave; w < 15; w += NW) wave_exclusive_scan_u32(pm.grp[w], G::NG + 1, lane);
        }
        __syncthreads();
        STAMP(17);
        for (int idx = tid; idx < ((prm.debug_skip & 32u) ? 0 : nseg); idx += G::T) {
            const int p = 31 - __clz(idx + 2);
            seg_static_eval(sh, n, p, (uint32_t)(idx + 2 - (1 << p)));
        }
        __syncthreads();
        STAMP(18);
        auto flush = [&pm](uint32_t idx, unsigned long long rc, unsigned long long bn, unsigned long long zr,
                           uint32_t hr) {
            atomicAdd(&pm.segacc[idx][0], rc);
            atomicAdd(&pm.segacc[idx][1], bn);
            atomicAdd(&pm.segacc[idx][2], zr);
            if (hr) atomicOr(&pm.segrun[idx], 1u);
        };
        if (prm.debug_skip & 8u) {
            // (timing ablation only)
        } else if (pnarrow && partitions_chunk_aligned<G>(n, max_p)) {
            // all orders in one walk (every full block, every probe); without a run of >= 4 zeros in the
            // block no partition can have one, so the zero-run costs are not needed.  Narrow sums: every segment total
            // stays below 2^32 (sum of u < 2^31, at most 35 bits of overhead per sample), so 32-bit LDS atomics on the
            // low words of the (zeroed) 64-bit accumulators suffice.
            // One atomic per accumulator and segment of neighbouring lanes (see seg_sum_u32): the lanes of a partition of
            // order p are (n >> p) / CH neighbours -- a power of two for every full block and every probe; other sizes
            // fall back to one atomic per lane.  Called by every lane of the wave (idle lanes pass zeros).
            const bool ablate_flush = (prm.debug_skip & 65536u) != 0u;
            const uint32_t chunks = n / (uint32_t)G::CH;  // chunks of the slot
            const bool pow2 = (chunks & (chunks - 1u)) == 0u && !(prm.debug_skip & 131072u);
            auto seg_flush = [&](int q, uint32_t idx, uint32_t rc, uint32_t bn, uint32_t zr, uint32_t hr, bool with_zr) {
                if (ablate_flush) return;
                const uint32_t lanes = chunks >> (q + 1);  // lanes per partition of this order (wave-uniform)
                if (pow2 && lanes >= 2u) {
                    const int lg = lanes >= 64u ? 6 : 31 - __clz((int)lanes);
                    const bool last = ((uint32_t)tid & ((1u << lg) - 1u)) == (1u << lg) - 1u;
                    rc = seg_sum_u32(rc, lg);
                    bn = seg_sum_u32(bn, lg);
                    if (with_zr) zr = seg_sum_u32(zr, lg);
                    if (last) {
                        atomicAdd(reinterpret_cast<uint32_t*>(&pm.segacc[idx][0]), rc);
                        atomicAdd(reinterpret_cast<uint32_t*>(&pm.segacc[idx][1]), bn);
                        if (with_zr) atomicAdd(reinterpret_cast<uint32_t*>(&pm.segacc[idx][2]), zr);
                    }
                } else {
                    atomicAdd(reinterpret_cast<uint32_t*>(&pm.segacc[idx][0]), rc);
                    atomicAdd(reinterpret_cast<uint32_t*>(&pm.segacc[idx][1]), bn);
                    if (with_zr) atomicAdd(reinterpret_cast<uint32_t*>(&pm.segacc[idx][2]), zr);
                }
                if (hr) atomicOr(&pm.segrun[idx], 1u);
            };
            auto flush32 = [&](int q, uint32_t idx, uint32_t rc, uint32_t bn, uint32_t zr, uint32_t hr) {
                seg_flush(q, idx, rc, bn, zr, hr, true);
            };
            auto flush32_nozr = [&](int q, uint32_t idx, uint32_t rc, uint32_t bn, uint32_t, uint32_t) {
                seg_flush(q, idx, rc, bn, 0u, 0u, false);
            };
            // (one queued chunk of partition_quick: any partition, any order per lane)
            auto flush_entry = [&pm, ablate_flush](uint32_t idx, uint32_t rc, uint32_t bn, uint32_t, uint32_t) {
                if (ablate_flush) return;
                atomicAdd(reinterpret_cast<uint32_t*>(&pm.segacc[idx][0]), rc);
                atomicAdd(reinterpret_cast<uint32_t*>(&pm.segacc[idx][1]), bn);
            };
            if (prm.zero_run && sh.best_hasrun) {
                partition_fused<G, true>(th, sh, max_p, flush32);
            } else if (prm.debug_skip & 32768u) {
                partition_fused<G, false>(th, sh, max_p, flush32_nozr);  // (A/B: the plain walk)
            } else {
                // no sample walk where the Rice parameter is provably constant over the chunk; the other (chunk, order)
                // pairs are queued and walked densely packed
                // (per wave: no workgroup barrier, no atomic -- a wave's queue is filled and drained by the wave itself)
                uint16_t* wq = &pm.queue[(tid >> 6) * 64 * G::MAXP];
                uint32_t queued = 0;  // wave-uniform
                partition_quick<G>(th, sh, max_p, flush32_nozr, [&](uint32_t entry, bool ambiguous) {
                    const unsigned long long m = __ballot(ambiguous);
                    if (ambiguous) wq[queued + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)entry;
                    queued += (uint32_t)__popcll(m);
                });
                STAMP(7);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS stores, before it reads them back
                for (uint32_t e = (uint32_t)(tid & 63); e < queued; e += 64u) partition_slow_entry<G>(sh, n, wq[e], flush_entry);
                STAMP(9);
            }
        } else {
            for (int p = 1; p <= max_p; ++p) {
                if (pnarrow) {
                    partition_pass<G, true>(th, sh, p, flush);
                } else {
                    partition_pass<G, false>(th, sh, p, flush);
                }
            }
        }
        STAMP(19);
        __syncthreads();
        STAMP(20);
        // Segment idx of order p has idx + 2 in [2^p, 2^(p+1)): walking j = idx + 2 in chunks of 64 gives every wave
        // from j = 64 on segments of ONE order -- one atomic per wave there instead of 64 on one address.
        for (int j0 = tid & ~63; j0 < nseg + 2; j0 += G::T) {  // wave-uniform trip count
            const int j = j0 + (tid & 63), idx = j - 2;
            const bool valid = j >= 2 && idx < nseg;
            const unsigned long long bits = valid ? seg_choose(sh, (uint32_t)idx, prm.zero_run) : 0ull;
            if (j0 >= 64) {
                const unsigned long long sum = wave_sum_u64(bits);
                if ((tid & 63) == 0) atomicAdd(&pm.pbits[31 - __clz(j0)], sum);
            } else if (valid) {
                atomicAdd(&pm.pbits[31 - __clz(j)], bits);
            }
        }
        __syncthreads();
    }
    if (tid == 0) finalize_plan(sh, n, prm.zero_run, max_p, &sh.plan);
    __syncthreads();
    // only the head and the partitions in use: the rest of the record is zero already (the plans are cleared per call)
    const int plan_words = (int)(offsetof(ChannelPlan, part_mode_k) + ((size_t)1 << sh.plan.partition_order) + 3) / 4;
    for (int i = tid; i < plan_words; i += G::T)
        reinterpret_cast<uint32_t*>(plan_out)[i] = reinterpret_cast<const uint32_t*>(&sh.plan)[i];
    STAMP(21);
    if constexpr (G::T == 1024) {
        if (fuse_idx >= 0) fused_emit<G>(sh, th, prm, fuse, fuse_idx, fuse_flag_byte, fuse_flag_value, tid STAMP_ARGS);  // uniform
    }
    STAMP(23);
#if defined(LACX_STAMPS) && LACX_STAMPS == 1
    // one wave per workgroup reports (a different one from workgroup to workgroup): with every wave adding its 24
    // sums to the same addresses the atomics themselves slowed every global load in the kernel down severalfold
    if ((tid & 63) == 0 && (tid >> 6) == (int)(blockIdx.x & 15u) && G::T == 1024) {
        stamp_acc[22] = __builtin_amdgcn_s_memrealtime() - stamp_rt0;
        for (int k = 0; k < 32; ++k) atomicAdd(&g_stamp_acc[k], stamp_acc[k]);
        atomicAdd(&g_stamp_acc[32], 1ull);
    }
#endif
}

// (At 127 VGPRs x 4 waves per SIMD an analysis workgroup fills the register files of its CU, so every workgroup of the
// streaming packer takes a whole CU away from the analysis: measured +25 us of kernel time per packer workgroup, hence
// the packer's small grid.  The compiler offers no way to cap this kernel at 120.)
template <class G>
__global__ __launch_bounds__(G::T, G::T == 64 ? 6 : 4) void k_analyze(BatchRef br, int probe_class, uint32_t one_block,
                                                  int which_base, const LpcSet* __restrict__ lpcs,
                                                  const uint32_t* __restrict__ need,
                                                  ChannelPlan* __restrict__ plans,
                                                  unsigned long long* __restrict__ t_first,
                                                  unsigned long long* __restrict__ t_last, FuseArgs fuse) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    // the earliest start, kept inverted (the word starts as zero like everything else the call clears)
    if (t_first && tid == 0) atomicMax(t_first, ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
    // Dense grids: consecutive workgroups are dealt round-robin to the 8 XCDs, so every launched workgroup
    // should be one that has work.  Whole-block class: the stream's workgroup w analyses the (w % per)-th needed slot of
    // its block w / per (per = the stream's channels; the streams of a set follow each other in the grid).  Probe class:
    // 12 slots per block, skipped unless the block is uncertain.  which_base != 0: the two extra workgroups of ONE block
    // (global block one_block) whose four channels are all needed.
    uint32_t blk;  // global block of the launch set
    int slot = -1;
    int which_in_block = -1;   // position of the slot among the block's needed whole-block slots
    uint32_t needed_slots = 0;
    StreamDesc sd;
    if (probe_class) {
        blk = blockIdx.x / 12u;
        sd = stream_of_block_uniform(br, blk);
        const int s = 4 + (int)(blockIdx.x % 12u);
        if ((need[blk] >> s) & 1u) slot = s;
    } else {
        uint32_t wsel;
        if (which_base) {
            blk = one_block;
            sd = stream_of_block_uniform(br, blk);
            wsel = blockIdx.x;
        } else {
            sd = stream_of_workgroup(br, blockIdx.x);
            const uint32_t per = sd.prm.channels == 2 ? 2u : 1u;
            uint32_t lblk;
            xcd_slot(blockIdx.x - sd.first_wg, per, (sd.prm.debug_skip & 512u) ? 0u : sd.prm.num_blocks, lblk, wsel);
            blk = sd.first_block + lblk;
        }
        int which = (int)wsel + which_base;
        which_in_block = which;
        uint32_t m = need[blk] & 0xFu;
        needed_slots = (uint32_t)__popc(m);
        while (m) {
            const int s = __ffs((int)m) - 1;
            if (which == 0) {
                slot = s;
                break;
            }
            --which;
            m &= m - 1u;
        }
    }
    if (slot < 0) return;  // uniform for the workgroup
    const AnalyzeParams prm = sd.prm;
    const uint32_t lblk = blk - sd.first_block;
    const SlotGeom g = slot_geom(prm, lblk, slot);
    const uint32_t n = g.n;
    const size_t sidx = (size_t)blk * kSlotsPerBlock + slot;
    const SlotSrc src = slot_src(prm, sd.left, sd.right, slot & 3);

    // Fused emit: only where the block's channel pair is already final, i.e. exactly `channels` whole-block slots are
    // needed (a small final block that is encoded both ways and compared afterwards is left to k_emit).
    long long fuse_idx = -1;
    bool flag_byte = false;
    if (fuse.slots && !probe_class && !which_base) {
        const uint32_t item = lblk * (uint32_t)prm.channels + (uint32_t)which_in_block;  // within the stream
        fuse_idx = (long long)prm.stream_base + item;
        flag_byte = prm.channels == 2 && prm.stereo_mode == 2 && which_in_block == 0;
        // the host excludes a small final block that may be encoded both ways and compared afterwards (fuse_items)
        if (item >= sd.fuse_items || needed_slots != (uint32_t)prm.channels) fuse_idx = -1;
    }
    analyze_slot<G>(smem_raw, prm, n, src, g.start, &lpcs[sidx], &plans[sidx], tid, fuse, fuse_idx, flag_byte,
                    (uint32_t)((slot & 3) >= 2 ? 1u : 0u));
    if (t_last && tid == 0) atomicMax(t_last, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

// ---------------------------------------------------------------------------------------------
// k_decide
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_decide(BatchRef br, int phase, BlockPlan* __restrict__ bplans,
                                               const uint32_t* __restrict__ need_probe,
                                               uint32_t* __restrict__ need_full,
                                               const ChannelPlan* __restrict__ plans) {
    // sixteen lanes per block, lane s reads slot s's size: one round of loads instead of twelve dependent cache misses
    const int tid = threadIdx.x, sub = tid & 15;
    const uint32_t blk = blockIdx.x * 4u + (uint32_t)(tid >> 4);
    // (only blocks of per-block-stereo streams are ever marked uncertain: k_stereo)
    if (br.table == nullptr && (br.single.prm.channels != 2 || br.single.prm.stereo_mode != 2)) return;  // uniform
    const bool live = blk < br.total_blocks;
    BlockPlan bp{};
    if (live) bp = bplans[blk];
    const bool mine = live && bp.uncertain &&
                      (phase == 1 ? need_probe[blk] != 0 : bp.frames <= (uint32_t)kFullCompareLimit);
    // phase 1: the 12 probe slots (4..15); phase 2: the whole-block slots (0..3)
    const bool take = mine && (phase == 1 ? sub >= 4 : sub < 4);
    const uint32_t bytes = take ? plans[(size_t)blk * kSlotsPerBlock + sub].payload_bytes : 0u;
    const bool is_ms = (sub & 3) >= 2;  // slot = window * 4 + channel, channels L R M S
    uint32_t lr = is_ms ? 0u : bytes, ms = is_ms ? bytes : 0u;  // sums of <= 12 sizes below 2^18: 32 bits
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        lr += (uint32_t)__shfl_xor((int)lr, d, 64);
        ms += (uint32_t)__shfl_xor((int)ms, d, 64);
    }
    if (!mine || sub != 0) return;
    bp.choose_ms = ms < lr;  // ref lac/encoder.cpp:347-353 (probes), :337-339 (small block)
    bplans[blk] = bp;
    if (phase == 1) need_full[blk] = bp.choose_ms ? 0xCu : 0x3u;
}

// ---------------------------------------------------------------------------------------------
// device-side emit: k_offsets + k_emit
// ---------------------------------------------------------------------------------------------
// One workgroup: byte size of every block's payload ([flag] + the two chosen channel blocks), exclusive
// prefix -> block_off[0..nb] (byte offset of the block in the result buffer), and the container's block table entries
// (frames, bytes).  In a set of several streams every stream's payload starts at its own region (StreamDesc::out_base):
// stream_pre[s] receives the prefix at the stream's first block and the offsets are re-based per stream.
__global__ __launch_bounds__(1024) void k_offsets(BatchRef br, const BlockPlan* __restrict__ bplans,
                                                   const ChannelPlan* __restrict__ plans,
                                                   unsigned long long* __restrict__ block_off,
                                                   uint32_t* __restrict__ table,
                                                   const unsigned long long* __restrict__ base_ptr,
                                                   unsigned long long* __restrict__ stream_pre) {
    __shared__ unsigned long long s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nb = br.total_blocks;
    const uint32_t per = (nb + 1023u) / 1024u;
    const uint32_t b0 = (uint32_t)tid * per;
    unsigned long long sum = 0;
    constexpr uint32_t kKeep = 8;  // block sizes kept in registers for the second pass (shards up to 8192 blocks)
    uint32_t kept[kKeep];
#pragma unroll
    for (uint32_t k = 0; k < kKeep; ++k) kept[k] = 0;
    for (uint32_t b = b0; b < b0 + per && b < nb; ++b) {
        const AnalyzeParams prm = stream_of_block(br, b).prm;
        const bool autost = prm.channels == 2 && prm.stereo_mode == 2;
        const ChannelPlan* p = plans + (size_t)b * kSlotsPerBlock;
        // every size the block could need, fetched at once (the kernel is one latency chain: no load waits for another)
        const uint32_t sl = p[CH_L].payload_bytes, sr = p[CH_R].payload_bytes, sm = p[CH_M].payload_bytes, ss = p[CH_S].payload_bytes;
        const BlockPlan bp = bplans[b];
        const bool ms = bp.choose_ms != 0;
        const uint32_t bytes = prm.channels == 1 ? sl : ((ms ? sm + ss : sl + sr) + (autost ? 1u : 0u));
        table[2 * b] = bp.frames;
        table[2 * b + 1] = bytes;
        if (b - b0 < kKeep) kept[b - b0] = bytes;
        sum += bytes;
    }
    const unsigned long long inc = wave_scan_add_u64(sum);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    unsigned long long base = base_ptr ? *base_ptr : 0ull;  // bytes of the chunks before this one
    for (int w = 0; w < wave; ++w) base += s_w[w];
    unsigned long long run = base + inc - sum;
    for (uint32_t b = b0; b < b0 + per && b < nb; ++b) {
        block_off[b] = run;
        uint32_t bytes = 0;
        if (b - b0 < kKeep) {
#pragma unroll
            for (uint32_t k = 0; k < kKeep; ++k) bytes = (b - b0 == k) ? kept[k] : bytes;
        } else {
            bytes = table[2 * b + 1];
        }
        run += bytes;
    }
    if (tid == 1023) block_off[nb] = base + inc;
    if (br.table != nullptr) {  // uniform
        __syncthreads();  // (block_off is global memory written by this workgroup: visible to it after the barrier)
        for (uint32_t sidx = (uint32_t)tid; sidx < br.nstreams; sidx += 1024u) stream_pre[sidx] = block_off[br.table[sidx].first_block];
        __syncthreads();
        for (uint32_t b = b0; b < b0 + per && b < nb; ++b) {
            const StreamDesc sd = stream_of_block(br, b);
            block_off[b] = block_off[b] - stream_pre[sd.pad] + sd.out_base;  // (pad = the stream's number in a table)
        }
    }
}

// One channel block of k_emit (workgroup-uniform control flow throughout).
template <class G>
__device__ __forceinline__ void emit_channel_block(EmitMem<G>& sh, int32_t* s_wx, const StreamDesc& sd,
                                                   const BlockPlan* __restrict__ bplans,
                                                   const ChannelPlan* __restrict__ plans,
                                                   const unsigned long long* __restrict__ block_off,
                                                   const uint32_t* __restrict__ table,
                                                   uint8_t* __restrict__ out,
                                                   uint32_t* __restrict__ err_flag, uint32_t blk, int which, int tid) {
    asm volatile("" : "+v"(tid));  // nothing derived from the thread index is hoisted out of the caller's loop (spills)
    const AnalyzeParams prm = sd.prm;
    const int32_t* __restrict__ L = sd.left;
    const int32_t* __restrict__ R = sd.right;
    const uint32_t lblk = blk - sd.first_block;
    const bool autost = prm.channels == 2 && prm.stereo_mode == 2;
    const bool ms = prm.channels == 2 && bplans[blk].choose_ms != 0;
    const int first_kind = prm.channels == 1 ? CH_L : (ms ? CH_M : CH_L);
    const int kind = which == 0 ? first_kind : (ms ? CH_S : CH_R);
    const ChannelPlan* plan = plans + (size_t)blk * kSlotsPerBlock + kind;
    const uint32_t n = block_frames(prm, lblk);
    unsigned long long off = block_off[blk] + (autost ? 1u : 0u);
    if (which == 1) off += plans[(size_t)blk * kSlotsPerBlock + first_kind].payload_bytes;
    // the destination is sized from an estimate: if this block does not fit, report it and write nothing
    if (block_off[blk] + table[2 * blk + 1] > sd.out_base + sd.out_cap) {
        if (tid == 0) atomicOr(err_flag, 2u);
        return;
    }
    if (which == 0 && autost && tid == 0) out[block_off[blk]] = ms ? 1 : 0;  // per-block flag (ref lac/encoder.cpp:363)

    Thread<G> th;
    thread_init(th, n, tid);
    stage_samples(th, sh, slot_src(prm, L, R, kind), (int64_t)lblk * kMaxBlock);
    emit_load_plan(sh, *plan, tid, G::T);
    if (tid == 0 && !plan->valid) sh.err = 1;
    __syncthreads();
    phase_r(th, sh, (int)sh.cand);
    emit_first_nonzero(th, sh);
    ScanRegs<G> sr;
    scan_pz_part1<G>(sh, tid, sr);
    const int32_t nxinc = scan_nx_part1<G>(sh, tid, s_wx);
    __syncthreads();
    scan_pz_part2<G>(sh, tid, sr);
    scan_nx_part2<G>(sh, tid, nxinc, s_wx, (int32_t)n);
    __syncthreads();
#ifdef LACX_STAMPS
    unsigned long long stamp_acc[40];
    unsigned long long stamp_prev = 0;
#endif
    emit_body<G>(sh, th, n, out, err_flag, [out, off](uint8_t** o) { *o = out + off; return true; }, tid, false, 0u STAMP_ARGS);
}

// k_emit: the bitstream of every channel block of the chunk that the fused emit has not produced (emitted[] == 0;
// all of them when emitted is null).  With a full grid every workgroup handles one channel block (XCD-aware mapping as
// in k_analyze); behind the fused emit the launcher uses a small grid that strides over the chunk, because then there
// is normally nothing left to do and a full grid of 1024-thread workgroups that exit at once is pure launch time.
template <class G>
__global__ __launch_bounds__(G::T) void k_emit(BatchRef br, uint32_t total_items, const BlockPlan* __restrict__ bplans,
                                               const ChannelPlan* __restrict__ plans,
                                               const unsigned long long* __restrict__ block_off,
                                               const uint32_t* __restrict__ table,
                                               uint8_t* __restrict__ out,
                                               uint32_t* __restrict__ err_flag,
                                               const uint32_t* __restrict__ emitted,
                                               const uint32_t* __restrict__ moved_total, uint32_t shard_items) {
    if (moved_total && *moved_total == shard_items) return;  // the streaming packer has moved everything (uniform)
    extern __shared__ __align__(16) unsigned char smem_raw[];
    EmitMem<G>& sh = *reinterpret_cast<EmitMem<G>*>(smem_raw);
    __shared__ int32_t s_wx[16];
    const int tid = threadIdx.x;
    // items = channel blocks of the launch set in stream-index order, numbered from the set's first stream index
    const uint32_t item0 = br.table ? br.table[0].prm.stream_base : br.single.prm.stream_base;
    for (uint32_t w = blockIdx.x; w < total_items; w += gridDim.x) {
        uint32_t item = w;
        if (br.table == nullptr && gridDim.x == total_items) {  // one stream, full grid: the XCD-aware mapping of k_analyze
            const uint32_t per = br.single.prm.channels == 2 ? 2u : 1u;
            uint32_t b, wsel;
            xcd_slot(w, per, total_items / per, b, wsel);
            item = b * per + wsel;
        }
        const StreamDesc sd = stream_of_item(br, item0 + item);
        const uint32_t per = sd.prm.channels == 2 ? 2u : 1u;
        const uint32_t local = item0 + item - sd.prm.stream_base;
        const uint32_t blk = sd.first_block + local / per, wsel = local % per;
        if (emitted && emitted[(size_t)item0 + item]) continue;  // uniform
        emit_channel_block<G>(sh, s_wx, sd, bplans, plans, block_off, table, out, err_flag, blk, (int)wsel, tid);
        __syncthreads();  // the LDS image is reused by the next channel block
    }
}

// `count` bytes from a staging slot (16-byte aligned, padded by 16 readable bytes) to dst (any alignment): 16-byte
// stores on 16-byte boundaries of the destination, the ragged head and tail bytewise.  NT cooperating threads (one
// 256-thread workgroup in k_pack, one wave in the streaming packer).
constexpr int kPackThreads = 256;
template <int NT = kPackThreads, int UNROLL = 4>
__device__ __forceinline__ void copy_slot_out(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t count,
                                              int tid) {
    const uint32_t* sw32 = reinterpret_cast<const uint32_t*>(src);
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u);
    const uint32_t head = mis ? (16u - mis < count ? 16u - mis : count) : 0u;
    const uint32_t nvec = (count - head) >> 4;
    if ((uint32_t)tid < head) dst[tid] = src[tid];
    {
        const uint32_t r = head & 3u, j0 = head >> 2;
        uint4* vdst = reinterpret_cast<uint4*>(dst + head);
#pragma unroll UNROLL
        for (uint32_t v = tid; v < nvec; v += NT) {
            const uint32_t j = j0 + 4u * v;
            uint32_t w[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) w[q] = sw32[j + q];  // j + 4 stays inside the slot's padding
            uint4 o;
            o.x = __builtin_amdgcn_alignbyte(w[1], w[0], r);
            o.y = __builtin_amdgcn_alignbyte(w[2], w[1], r);
            o.z = __builtin_amdgcn_alignbyte(w[3], w[2], r);
            o.w = __builtin_amdgcn_alignbyte(w[4], w[3], r);
            vdst[v] = o;
        }
    }
    const uint32_t t0 = head + (nvec << 4);
    if (t0 + (uint32_t)tid < count) dst[t0 + tid] = src[t0 + tid];
}

// k_stream_out: the streaming packer (see "Fused emit + streaming packer" above).  Every WAVE is a packer of its own:
// wave u of U moves the stream indices u, u + U, ... and keeps its own running byte offset by summing the size records
// of every index in order (64 per load round); no barrier, no shared memory.  The waves come as 1024-thread workgroups
// because of where they run: an analysis workgroup fills the register files of its CU, so a packer workgroup takes a
// whole CU away from the analysis however small it is -- sixteen packer waves on one CU cost the analysis one CU, eight
// 256-thread workgroups cost it eight.  One CU moves about 16 GB/s into pinned host memory however many stores it keeps
// in flight, so the 10 min stream's 72.6 MB in 2.5 ms need two.  Measured (ms per step: 16/48 music, 16/48 mixed,
// 24/96 mixed): 8 x 256 threads 3.15 / 4.15 / 8.33; 1 x 1024 4.02 (packer too slow) / 3.99 / -; 2 x 1024 3.12 / 4.02 /
// 8.15; 3 x 1024 3.27 / 3.98 / 8.22; 4 x 1024 3.29 / 3.96 / 8.17.  total: fusable stream indices of the shard.
#ifndef LACX_STREAM_UNROLL
#define LACX_STREAM_UNROLL 8
#endif
constexpr int kStreamGrid = 2;        // packing straight into pinned host memory (PCIe-bound: see above)
constexpr int kStreamGridDevice = 3;  // packing into device memory that a copy engine drains: three CUs keep up with the analysis
constexpr uint32_t kRangeItems = 256;  // stream indices per progress range (about 5 MB of 16-bit music)
constexpr int kStreamThreads = 1024;
constexpr unsigned long long kStreamTimeoutTicks = 2000000ull;  // 20 ms of the 100 MHz clock without the awaited record
__global__ __launch_bounds__(kStreamThreads) void k_stream_out(BatchRef br, uint32_t total, int nap,
                                                               const uint16_t* __restrict__ item_stream,
                                                               const unsigned long long* __restrict__ size_rec,
                                                               const unsigned long long* __restrict__ ready_rec,
                                                               const uint8_t* __restrict__ slots, unsigned long long slot_stride,
                                                               uint8_t* __restrict__ out,
                                                               uint32_t* __restrict__ packed, uint32_t* __restrict__ err_flag,
                                                               uint32_t* __restrict__ moved_total, uint32_t* __restrict__ gave_up,
                                                               RangeProgress rp) {
    // total: stream indices of the set (all streams).  In a set of several streams every stream's payload has its own
    // region of the result buffer and its own running offset; item_stream[i] = the stream of index i (null: one stream).
    const int lane = threadIdx.x & 63;
    const uint32_t unit = blockIdx.x * (uint32_t)(kStreamThreads / 64) + (threadIdx.x >> 6);
    const uint32_t units = gridDim.x * (uint32_t)(kStreamThreads / 64);
    unsigned long long running = 0;  // bytes of the stream indices [first index of the current stream, summed)
    uint32_t summed = br.table ? 0u : br.single.prm.stream_base;
    uint32_t cur_first = summed;     // first stream index of the stream `running` belongs to
    uint32_t moved = 0;  // stream indices this wave has put in place
    // Progress for the host (one stream, device destination that a copy engine drains while the analysis goes on): the
    // indices come in ranges of kRangeItems; a wave's indices are `units` apart, so it has a handful per range.  When it
    // leaves a range it writes its stores back to memory -- a copy engine does not look into the L2 -- and adds its count
    // to the range's; the wave that completes the count publishes the range's end offset (left by the wave that moved
    // the range's last index) to the host.
    uint32_t pend_range = 0xFFFFFFFFu, pend_count = 0;
    auto flush_progress = [&]() {
        if (pend_count == 0u) return;  // wave-uniform
        if (rp.fence_mode == 0u) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: s_waitcnt + L2 write-back
        else if (rp.fence_mode == 1u) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const uint32_t r = pend_range;
            const uint32_t last = (r + 1u) * kRangeItems - 1u < rp.fuse_total - 1u ? (r + 1u) * kRangeItems - 1u : rp.fuse_total - 1u;
            const uint32_t in_range = last - r * kRangeItems + 1u;
            const uint32_t before = __hip_atomic_fetch_add(&rp.range_cnt[r], pend_count, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (before + pend_count == in_range) {
                const unsigned long long end = __hip_atomic_load(&rp.range_end[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&rp.host_end[r], end + 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        pend_count = 0;
    };
    for (uint32_t i = unit; i < total; i += units) {
        const StreamDesc sd = br.table ? br.table[item_stream[i]] : br.single;
        const uint32_t first = sd.prm.stream_base;
        if (i - first >= sd.fuse_items) continue;  // left to k_emit (a small final block that is encoded both ways)
        if (first != cur_first) {  // a new stream: its offsets start over in its own region
            cur_first = first;
            summed = first;
            running = 0;
        }
        bool alive = true;
        unsigned long long mine = 0;
        // sizes of [summed, i], 64 records per round; the last one is this index's own
        while (alive && summed <= i) {
            const uint32_t j = summed + (uint32_t)lane;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned long long v;
            for (;;) {
                v = (j <= i) ? rec_load(&size_rec[j]) : kRecValid;
                if (__ballot((v & kRecValid) == 0ull) == 0ull) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > kStreamTimeoutTicks) {
                    alive = false;
                    break;
                }
                for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(16);
            }
            if (!alive) break;
            const uint32_t cnt = (i - summed + 1u) < 64u ? (i - summed + 1u) : 64u;
            const bool last_round = summed + cnt == i + 1u;
            // everything but this index's own record goes into the running offset
            const bool take = (uint32_t)lane < cnt && !(last_round && (uint32_t)lane == cnt - 1u);
            running += wave_sum_u64(take ? (v & kRecBytesMask) : 0ull);
            if (last_round) mine = __shfl(v, (int)cnt - 1, 64);
            summed += cnt;
        }
        unsigned long long ready = 0;
        if (alive) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                ready = rec_load(&ready_rec[i]);
                if (ready != 0ull) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > kStreamTimeoutTicks) {
                    alive = false;
                    break;
                }
                for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(16);
            }
        }
        if (!alive) {  // wave-uniform: a producer went missing; k_pack / k_emit move what is left
            if (lane == 0) atomicAdd(gave_up, 1u);
            break;
        }
        const unsigned long long off = running, rec = mine;
        running += mine & kRecBytesMask;  // this index is accounted for whatever happens to its bytes
        if (ready == 1ull) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const unsigned long long bytes = rec & kRecBytesMask;
            const bool flag_byte = (rec & kRecFlag) != 0ull;
            if (off + bytes > sd.out_cap) {  // the destination was sized from an estimate: report, write nothing
                if (lane == 0) atomicOr(err_flag, 2u);
            } else {
                uint8_t* dst = out + sd.out_base + off;
                const uint32_t fb = flag_byte ? 1u : 0u;
                if (flag_byte && lane == 0) dst[0] = (rec & kRecMs) ? 1 : 0;  // per-block flag (ref lac/encoder.cpp:363)
                copy_slot_out<64, LACX_STREAM_UNROLL>(slots + (unsigned long long)i * slot_stride, dst + fb, (uint32_t)bytes - fb, lane);
                if (lane == 0) packed[i] = 1u;
                ++moved;
            }
        } else if (lane == 0) {
            atomicOr(err_flag, 4u);  // nothing came from the analysis kernel for this index: its bytes arrive later (k_emit)
        }
        // Progress for the host (one stream, device destination): see flush_progress.
        if (rp.host_end) {
            const uint32_t r = i / kRangeItems;
            if (r != pend_range) {
                flush_progress();
                pend_range = r;
            }
            const uint32_t last_of_range = (r + 1u) * kRangeItems - 1u < rp.fuse_total - 1u ? (r + 1u) * kRangeItems - 1u : rp.fuse_total - 1u;
            if (i == last_of_range && lane == 0) __hip_atomic_store(&rp.range_end[r], running, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ++pend_count;
        }
    }
    flush_progress();
    // k_pack / k_emit behind this kernel return at once when every channel block of the shard was moved here
    if (lane == 0 && moved) atomicAdd(moved_total, moved);
}

// k_pack: copies the channel blocks that the fused emit has written to their staging slots to their place in the shard
// payload (usually pinned host memory behind PCIe), in 16-byte stores on 16-byte boundaries of the destination; the
// ragged head and tail go bytewise.  Pure data movement at the pace of the PCIe link, so the grid is deliberately
// small (kPackGrid workgroups striding over the channel blocks): a grid of one workgroup per channel block would fill
// every wave slot of the chip with waves that wait for PCIe and lock the next chunk's analysis kernel out.
constexpr int kPackGrid = 96;
__global__ __launch_bounds__(kPackThreads) void k_pack(BatchRef br, uint32_t total_items, const BlockPlan* __restrict__ bplans,
                                                       const ChannelPlan* __restrict__ plans,
                                                       const unsigned long long* __restrict__ block_off,
                                                       const uint32_t* __restrict__ table,
                                                       uint8_t* __restrict__ out,
                                                       uint32_t* __restrict__ err_flag, const uint8_t* __restrict__ slots,
                                                       unsigned long long slot_stride,
                                                       const uint32_t* __restrict__ emitted,
                                                       const uint32_t* __restrict__ packed,
                                                       const uint32_t* __restrict__ moved_total, uint32_t shard_items,
                                                       uint32_t* __restrict__ repacked) {
    if (moved_total && *moved_total == shard_items) return;  // the streaming packer has moved everything (uniform)
    const int tid = threadIdx.x;
    const uint32_t item0 = br.table ? br.table[0].prm.stream_base : br.single.prm.stream_base;
    for (uint32_t work = blockIdx.x; work < total_items; work += gridDim.x) {
        const size_t idx = (size_t)item0 + work;
        if (emitted[idx] != 2u || (packed && packed[idx])) continue;  // not in its slot (k_emit's job) / moved by the packer
        const StreamDesc sd = stream_of_item(br, (uint32_t)idx);
        const AnalyzeParams prm = sd.prm;
        const uint32_t per = prm.channels == 2 ? 2u : 1u;
        const uint32_t local = (uint32_t)idx - prm.stream_base;
        const uint32_t blk = sd.first_block + local / per;
        const int which = (int)(local % per);
        const bool autost = prm.channels == 2 && prm.stereo_mode == 2;
        const bool ms = prm.channels == 2 && bplans[blk].choose_ms != 0;
        const int first_kind = prm.channels == 1 ? CH_L : (ms ? CH_M : CH_L);
        const int kind = which == 0 ? first_kind : (ms ? CH_S : CH_R);
        const uint32_t count = plans[(size_t)blk * kSlotsPerBlock + kind].payload_bytes;
        unsigned long long off = block_off[blk] + (autost ? 1u : 0u);
        if (which == 1) off += plans[(size_t)blk * kSlotsPerBlock + first_kind].payload_bytes;
        // the destination is sized from an estimate: if this block does not fit, report it and write nothing
        if (block_off[blk] + table[2 * blk + 1] > sd.out_base + sd.out_cap) {
            if (tid == 0) atomicOr(err_flag, 2u);
            continue;
        }
        if (which == 0 && autost && tid == 0) out[block_off[blk]] = ms ? 1 : 0;  // per-block flag (ref lac/encoder.cpp:363)
        copy_slot_out(slots + idx * slot_stride, out + off, count, tid);
        if (tid == 0 && repacked) atomicAdd(repacked, 1u);
    }
}

// k_gather: see GatherList (kernels.h).
__global__ __launch_bounds__(256) void k_gather(GatherList g) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x, gsz = gridDim.x * 256u;
    for (int k = 0; k < g.n; ++k) {
        const uint32_t* __restrict__ s = static_cast<const uint32_t*>(g.src[k]);
        uint32_t* __restrict__ d = static_cast<uint32_t*>(g.dst[k]);
        for (uint32_t i = gid; i < g.words[k]; i += gsz) d[i] = s[i];
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launcher
// ---------------------------------------------------------------------------------------------
using GFull = Geo<16, 1024>;
using GProbe = Geo<4, 64>;
constexpr int kMaxDevices = 64;

size_t analyze_smem_bytes_full() { return sizeof(Smem<GFull>); }

int debug_read_stamps(unsigned long long* out32) {
#ifdef LACX_STAMPS
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_stamp_acc), sizeof(unsigned long long) * 40) != hipSuccess) return 0;
    unsigned long long zero[40] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_acc), zero, sizeof(zero));
    return 1;
#else
    (void)out32;
    return 0;
#endif
}
size_t analyze_smem_bytes_probe() { return sizeof(Smem<GProbe>); }

// The opt-in to more than 64 KiB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize) applies to the device
// that is current when it is set, and one process may drive several devices (one encoder per lacx_config.device):
// the state is kept per device ordinal, under a mutex (first launches of two encoders may come from two host
// threads), and only successes are remembered -- a transient failure is retried by the next call.
static hipError_t ensure_kernel_attrs() {
    static std::mutex mu;
    static bool done[kMaxDevices] = {};
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    if (done[dev]) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_analyze<GFull>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem<GFull>));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_levinson), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)sizeof(LevMem));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_analyze<GProbe>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem<GProbe>));
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_emit<GFull>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(EmitMem<GFull>));
    if (e == hipSuccess) done[dev] = true;
    return e;
}

hipError_t launch_emit(const LaunchSet& ls, const DeviceWorkspace& ws, uint8_t* out,
                       const unsigned long long* base_ptr, hipEvent_t wait_before_offsets,
                       hipEvent_t offsets_done, hipStream_t stream, bool skip_emitted, const uint32_t* moved_total,
                       uint32_t shard_items, hipEvent_t wait_before_pack, uint32_t* repacked) {
    const hipError_t attr_err = ensure_kernel_attrs();
    if (attr_err != hipSuccess) return attr_err;
    const uint32_t nb = ls.br.total_blocks;
    if (nb == 0) return hipSuccess;
    if (wait_before_offsets) {
        const hipError_t we = hipStreamWaitEvent(stream, wait_before_offsets, 0);
        if (we != hipSuccess) return we;
    }
    hipLaunchKernelGGL(k_offsets, dim3(1), dim3(1024), 0, stream, ls.br, ws.bplans, ws.plans, ws.block_off, ws.table,
                       base_ptr, ws.stream_pre);
    if (offsets_done) {
        const hipError_t re = hipEventRecord(offsets_done, stream);
        if (re != hipSuccess) return re;
    }
    if (wait_before_pack) {  // the streaming packer has to be through before anybody looks at what it left behind
        const hipError_t we = hipStreamWaitEvent(stream, wait_before_pack, 0);
        if (we != hipSuccess) return we;
    }
    const uint32_t work = ls.total_items;
    if (skip_emitted && ws.slots) {
        hipLaunchKernelGGL(k_pack, dim3(work < (uint32_t)kPackGrid ? work : (uint32_t)kPackGrid), dim3(kPackThreads), 0, stream, ls.br, work,
                           ws.bplans, ws.plans, ws.block_off, (const uint32_t*)ws.table, out, ws.err_flag, (const uint8_t*)ws.slots,
                           ws.slot_stride, (const uint32_t*)ws.emitted, (const uint32_t*)ws.packed, moved_total, shard_items, repacked);
    }
    const bool leftovers_only = skip_emitted && ws.slots;  // behind the fused emit
    hipLaunchKernelGGL(k_emit<GFull>, dim3(leftovers_only && work > 64u ? 64u : work), dim3(GFull::T),
                       sizeof(EmitMem<GFull>), stream, ls.br, work, ws.bplans, ws.plans, ws.block_off, (const uint32_t*)ws.table,
                       out, ws.err_flag, skip_emitted ? (const uint32_t*)ws.emitted : (const uint32_t*)nullptr,
                       leftovers_only ? moved_total : (const uint32_t*)nullptr, shard_items);
    return hipGetLastError();
}

hipError_t launch_gather(const GatherList& g, hipStream_t stream) {
    if (g.n == 0) return hipSuccess;
    uint32_t most = 0;
    for (int k = 0; k < g.n; ++k) most = g.words[k] > most ? g.words[k] : most;
    const uint32_t grid = most <= 256u ? 1u : (most + 255u) / 256u > 32u ? 32u : (most + 255u) / 256u;
    hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, stream, g);
    return hipGetLastError();
}

static_assert(kRangeItems == kPackerRangeItems, "host and device agree on the range size");
hipError_t launch_stream_out(const LaunchSet& ls, const DeviceWorkspace& ws, uint8_t* out, uint32_t* counters,
                             hipStream_t stream, const RangeProgress& rp) {
    if (ls.total_items == 0) return hipSuccess;
    int nap = 1, grid = rp.host_end ? kStreamGridDevice : kStreamGrid;  // tuning knobs
    if (const char* v = std::getenv("LACX_PACK_NAP")) nap = std::atoi(v) > 0 ? std::atoi(v) : 1;
    if (const char* v = std::getenv("LACX_PACK_GRID")) grid = std::atoi(v) > 0 ? std::atoi(v) : grid;
    // counters: [0] error flags, [1] channel blocks put in place, [2] packer waves that gave up waiting
    hipLaunchKernelGGL(k_stream_out, dim3((uint32_t)grid), dim3(kStreamThreads), 0, stream, ls.br, ls.total_items, nap, ls.item_stream,
                       (const unsigned long long*)ws.size_rec, (const unsigned long long*)ws.ready_rec, (const uint8_t*)ws.slots,
                       ws.slot_stride, out, ws.packed, counters, counters + 1, counters + 2, rp);
    return hipGetLastError();
}

hipError_t launch_analysis(const LaunchSet& ls, const DeviceWorkspace& ws, hipStream_t stream, hipEvent_t* ev,
                           const FuseArgs* fuse, hipEvent_t wait_before_full) {
    const FuseArgs fa = fuse ? *fuse : FuseArgs{};
    hipError_t e = ensure_kernel_attrs();
    if (e != hipSuccess) return e;
    const BatchRef& br = ls.br;
    const uint32_t nb = br.total_blocks;
    if (nb == 0) return hipSuccess;
    if (ev) (void)hipEventRecord(ev[0], stream);
    e = hipMemsetAsync(ws.plans, 0, sizeof(ChannelPlan) * (size_t)nb * kSlotsPerBlock, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_ingest, dim3(nb * 4u), dim3(kIngestThreads), 0, stream, br, ws.sums, ws.badidx, ws.acorr);
    hipLaunchKernelGGL(k_stereo, dim3((nb + 3) / 4), dim3(64), 0, stream, br, ws.sums, ws.badidx, ws.bplans,
                       ws.need_probe, ws.need_full);
    hipLaunchKernelGGL(k_levinson, dim3((nb * kSlotsPerBlock + kLevThreads - 1) / kLevThreads), dim3(kLevThreads),
                       sizeof(LevMem), stream, br, ws.acorr, ws.need_probe, ws.lpcs);
    if (ev) (void)hipEventRecord(ev[1], stream);
    // per-block stereo in any stream of the set: probes + decision; a final block of <= 4096 frames of such a stream can
    // need all four channels (full LR-vs-MS comparison, ref lac/encoder.cpp:336-340)
    bool any_auto = false, any_both = false;
    uint32_t total_wg = 0;
    for (uint32_t i = 0; i < ls.nstreams; ++i) {
        const AnalyzeParams& p = ls.streams[i].prm;
        const bool autost = p.channels == 2 && p.stereo_mode == 2;
        any_auto = any_auto || autost;
        any_both = any_both || (autost && p.frames - (uint64_t)(p.num_blocks - 1) * kMaxBlock <= (uint64_t)kFullCompareLimit);
        total_wg += p.num_blocks * (p.channels == 2 ? 2u : 1u);
    }
    if (any_auto) {
        hipLaunchKernelGGL(k_analyze<GProbe>, dim3(nb * 12u), dim3(GProbe::T), sizeof(Smem<GProbe>), stream, br, 1, 0u, 0,
                           ws.lpcs, ws.need_probe, ws.plans, (unsigned long long*)nullptr, (unsigned long long*)nullptr, FuseArgs{});
        hipLaunchKernelGGL(k_decide, dim3((nb + 3) / 4), dim3(64), 0, stream, br, 1, ws.bplans, ws.need_probe,
                           ws.need_full, ws.plans);
    }
    if (ev) (void)hipEventRecord(ev[2], stream);
    if (wait_before_full) {
        const hipError_t we = hipStreamWaitEvent(stream, wait_before_full, 0);
        if (we != hipSuccess) return we;
    }
    hipLaunchKernelGGL(k_analyze<GFull>, dim3(total_wg), dim3(GFull::T), sizeof(Smem<GFull>), stream, br, 0, 0u, 0,
                       ws.lpcs, ws.need_full, ws.plans, ws.t_first, ws.t_last, fa);
    if (any_both) {  // the 3rd and 4th slots of such a final block: a two-workgroup launch each
        for (uint32_t i = 0; i < ls.nstreams; ++i) {
            const StreamDesc& sd = ls.streams[i];
            const AnalyzeParams& p = sd.prm;
            if (p.channels == 2 && p.stereo_mode == 2 && p.frames - (uint64_t)(p.num_blocks - 1) * kMaxBlock <= (uint64_t)kFullCompareLimit)
                hipLaunchKernelGGL(k_analyze<GFull>, dim3(2), dim3(GFull::T), sizeof(Smem<GFull>), stream, br, 0,
                                   sd.first_block + p.num_blocks - 1u, 2, ws.lpcs, ws.need_full, ws.plans,
                                   (unsigned long long*)nullptr, (unsigned long long*)nullptr, FuseArgs{});
        }
    }
    if (ev) (void)hipEventRecord(ev[3], stream);
    if (any_both) {  // phase 2 only concerns such final blocks
        hipLaunchKernelGGL(k_decide, dim3((nb + 3) / 4), dim3(64), 0, stream, br, 2, ws.bplans, ws.need_probe,
                           ws.need_full, ws.plans);
    }
    if (ev) (void)hipEventRecord(ev[4], stream);
    return hipGetLastError();
}

}  // namespace lacx
