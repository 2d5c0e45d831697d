// k_front.hip -- the kernels in front of the whole-block analysis (all launches asynchronous on one stream):
//   k_ingest    one workgroup per (block, channel): sample-range validation, the proxy sums of
//               estimate_stereo_mode, exact 13-lag int64 autocorrelation of the whole block and of the
//               3 probe windows                             (ref lac/encoder.cpp:82-102,126-178; lpc.cpp:80-96)
//   k_stereo    sixteen lanes per block: LR/MS estimate -> BlockPlan, need masks (ref lac/encoder.cpp:179-196)
//   k_levinson  one lane per slot: Levinson-Durbin in software x87 extended precision -> Q15 sets
//                                                           (ref lpc.cpp:98-186)
//   k_decide(1) probes -> LR/MS choice, marks the two whole-block slots still to be analysed
//   k_decide(2) small-block full comparison (ref lac/encoder.cpp:336-340), final BlockPlan (only for such a block)
// Launched by launch_analysis (k_analyze.hip).
#include <hip/hip_runtime.h>

#include "kernels_internal.h"
#include "x87.h"

namespace lacx {

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

__device__ __forceinline__ void stereo_block(const AnalyzeParams& prm, uint32_t blk, uint32_t nb, bool live, int tid,
                                             const unsigned long long* __restrict__ sums, const uint32_t* __restrict__ badidx,
                                             BlockPlan* __restrict__ bplans, uint32_t* __restrict__ need_probe,
                                             uint32_t* __restrict__ need_full);

// The workgroup is done with its (block, channel): count it; the last of the block's four makes the block's stereo estimate
// (front_ctr, kernels_internal.h).  Thread 0 has written the workgroup's sums / first bad index just before, with
// agent-scope (write-through) stores: it waits for them, then counts; the last workgroup reads them with agent-scope
// loads.  No fence on either side (a release fence per workgroup writes the whole L2 back: measured +0.1 ms per kernel).
__device__ __forceinline__ void ingest_done(uint32_t* __restrict__ front_ctr, unsigned int* s_last, const AnalyzeParams& prm,
                                            uint32_t blk, uint32_t nb, const unsigned long long* __restrict__ sums,
                                            const uint32_t* __restrict__ badidx, BlockPlan* __restrict__ bplans,
                                            uint32_t* __restrict__ need_probe, uint32_t* __restrict__ need_full) {
    if (!front_ctr) return;  // uniform
    const int tid = threadIdx.x;
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *s_last = atomicAdd(&front_ctr[(size_t)blk * 2], 1u) == 3u ? 1u : 0u;
    }
    __syncthreads();
    if (*s_last == 0u || tid >= 64) return;  // (uniform per wave)
    if (tid == 0) front_ctr[(size_t)blk * 2] = 0u;  // nobody else looks at it any more in this call
    stereo_block(prm, blk, nb, tid < 16, tid, sums, badidx, bplans, need_probe, need_full);
}

__global__ __launch_bounds__(kIngestThreads, 5) void k_ingest(BatchRef br, unsigned long long* __restrict__ sums,
                                                           uint32_t* __restrict__ badidx,
                                                           int64_t* __restrict__ acorr, uint32_t* __restrict__ front_ctr,
                                                           BlockPlan* __restrict__ bplans, uint32_t* __restrict__ need_probe,
                                                           uint32_t* __restrict__ need_full) {
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
    const uint32_t nb = block_frames(prm, lblk);
    if (!used && !validate) {  // uniform
        if (ch < 2 && threadIdx.x == 0) agent_store(&badidx[blk * 2 + ch], 0xFFFFFFFFu);
        ingest_done(front_ctr, &s_bad, prm, blk, nb, sums, badidx, bplans, need_probe, need_full);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
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
            agent_store(&sums[(size_t)blk * 12 + ch], s_sum[0]);
            agent_store(&sums[(size_t)blk * 12 + 4 + ch], s_sum[1]);
            agent_store(&sums[(size_t)blk * 12 + 8 + ch], s_sum[2]);
        }
        if (validate || ch < 2) agent_store(&badidx[blk * 2 + ch], s_bad);
    }
    __syncthreads();  // (thread 0 has read s_bad: the word is reused)
    ingest_done(front_ctr, &s_bad, prm, blk, nb, sums, badidx, bplans, need_probe, need_full);
}

// Sixteen lanes per block, four blocks per wave: lanes 0..11 of a block turn one of its 12 proxy sums into bits (one
// 64-bit division each instead of a chain of twelve), lane 0 of the block decides.
constexpr int kStereoLanes = 16;
// The estimate of ONE block by the sixteen lanes of a lane group (tid = lane in the wave; every lane of the wave calls, so
// that the shuffles find their sources; `live` = this group has a block).
__device__ __forceinline__ void stereo_block(const AnalyzeParams& prm, uint32_t blk, uint32_t nb, bool live, int tid,
                                             const unsigned long long* __restrict__ sums, const uint32_t* __restrict__ badidx,
                                             BlockPlan* __restrict__ bplans, uint32_t* __restrict__ need_probe,
                                             uint32_t* __restrict__ need_full) {
    const int sub = tid & (kStereoLanes - 1), grp = tid & ~(kStereoLanes - 1);
    const bool stereo = prm.channels == 2;
    const bool est = stereo && prm.stereo_mode == 2;
    // estimate_channel_proxy_cost: ref lac/encoder.cpp:114-124 -- sums[blk][kind * 4 + channel], kind = raw, diff, anti
    uint64_t bits = 0;
    if (est && live && sub < 12) bits = approx_rice_bits(agent_load(&sums[(size_t)blk * 12 + sub]), nb);
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
    const uint32_t badl = agent_load(&badidx[blk * 2]), badr = stereo ? agent_load(&badidx[blk * 2 + 1]) : 0xFFFFFFFFu;
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
        } else if (agent_load(&sums[(size_t)blk * 12]) == 0ull && agent_load(&sums[(size_t)blk * 12 + 1]) == 0ull) {
            // Both raw sums are zero: every sample of the block is zero (digital silence), so are mid and side, the
            // twelve probe encodes are twelve times the same bytes, the comparison ties and left/right stays
            // (ref lac/encoder.cpp:347-353: mid/side only when strictly smaller).  No probes, no Levinson for them.
            nfull = 0x3u;
            bp.choose_ms = 0;
        } else {
            nprobe = 0xFFF0u;  // 12 probe slots; the whole-block pair is picked by k_decide phase 1
        }
    }
    bplans[blk] = bp;
    need_probe[blk] = nprobe;
    need_full[blk] = nfull;
}

__global__ __launch_bounds__(64) void k_stereo(BatchRef br, const unsigned long long* __restrict__ sums,
                                               const uint32_t* __restrict__ badidx, BlockPlan* __restrict__ bplans,
                                               uint32_t* __restrict__ need_probe, uint32_t* __restrict__ need_full) {
    const int tid = threadIdx.x;
    const uint32_t blk = blockIdx.x * (64 / kStereoLanes) + (uint32_t)(tid / kStereoLanes);
    const bool live = blk < br.total_blocks;  // every lane stays for the shuffles
    const StreamDesc sd = stream_of_block(br, live ? blk : 0u);
    const uint32_t nb = live ? block_frames(sd.prm, blk - sd.first_block) : 0u;
    stereo_block(sd.prm, blk, nb, live, tid, sums, badidx, bplans, need_probe, need_full);
}

// ---------------------------------------------------------------------------------------------
// k_levinson: one lane per slot that needs it
// ---------------------------------------------------------------------------------------------

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

hipError_t set_kernel_attrs_front() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_levinson), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)sizeof(LevMem));
}

}  // namespace lacx
