// k_analyze.hip -- the whole-block / probe analysis kernel and the launcher of the analysis pipeline.
//   k_analyze<4,64>     one wave per probe slot of an "uncertain" block (ref lac/encoder.cpp:341-354)
//   k_analyze<16,1024>  one 1024-thread workgroup per needed whole-block slot
//                                                           (ref block/encoder.cpp:313-552)
//                       + the bit emit of its channel block into a staging slot (ref block/encoder.cpp:554-838)
// Pipeline per launch set (launch_analysis): k_ingest, k_stereo, k_levinson (k_front.hip), k_analyze<4,64>, k_decide(1),
// k_analyze<16,1024> [+ two workgroups per small final block, k_decide(2)]; beside the whole-block kernel, on its own
// stream, the streaming packer k_stream_out; behind it k_offsets, k_pack, k_emit, k_gather (k_emit.hip).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>

#include "emit_device.h"
#include "kernels_internal.h"

namespace lacx {

#ifdef LACX_STAMPS
__device__ unsigned long long g_stamp_acc[40];
#endif

// Test hooks and timing ablations (AnalyzeParams::debug_skip, from LACX_DEBUG_SKIP) exist only in the diagnostic library
// (liblacx_hooks.so, -DLACX_TEST_HOOKS): bits 10 / 11 / 13 force the repair paths of the fused emit for the parity tests,
// the other bits switch phases off for timing experiments.  The production kernel carries none of them.
#ifdef LACX_TEST_HOOKS
#define LACX_HOOK(prm, bits) (((prm).debug_skip & (bits)) != 0u)
#else
#define LACX_HOOK(prm, bits) false
#endif

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
// A slot whose bitstream has been stored but not announced yet (persistent workgroups).  Lives in LDS: only thread 0 ever
// acts on it, and a struct handed down by pointer through the analysis is kept in scratch memory by the compiler (two
// stores and four loads per slot and thread, a quarter of the kernel's scratch traffic).
struct PendingSlot {
    long long idx;
    uint32_t done, silent;
};

// idx: stream index of this channel block; flag_byte: the block's LR/MS flag byte precedes this channel block.
template <class G>
__device__ __forceinline__ void fused_emit(Smem<G>& sh, Thread<G>& th, const AnalyzeParams& prm, const FuseArgs& fa,
                                           const long long idx, const bool flag_byte, const uint32_t flag_value,
                                           const int tid, PendingSlot* defer, const uint32_t tmpl_key,
                                           const uint32_t* plan_stored, const int plan_words STAMP_PARAMS) {
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
                      (LACX_HOOK(prm, 1024u) && (idx % 5 == 3));
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
                            LACX_HOOK(prm, 2048u), (uint32_t)fa.slot_stride STAMP_ARGS);
    }
    // A silent slot, and nobody has left its finished channel block behind yet (tmpl_key != 0): this workgroup does, if it
    // is the first to ask.  Plain stores, drained by every wave, the workgroup meets, one release store of the key.
    if (tmpl_key != 0u && done && sh.payload_bytes <= kSilentBytes) {  // uniform
        SilentTemplate* t = fa.silent;
        if (tid == 0) sh.tmpl_state = atomicCAS(&t->state, 0u, 1u);
        __syncthreads();
        if (sh.tmpl_state == 0u) {  // uniform
            const uint32_t nv = (sh.payload_bytes + 15u) >> 4;
            const uint32_t* tw = sh.xp.o.obits;  // (the whole bitstream is in the one tile: kSilentBytes < the tile)
            for (uint32_t v = tid; v < nv; v += G::T) {
                uint4 o;
                o.x = __builtin_bswap32(tw[4u * v]);
                o.y = __builtin_bswap32(tw[4u * v + 1u]);
                o.z = __builtin_bswap32(tw[4u * v + 2u]);
                o.w = __builtin_bswap32(tw[4u * v + 3u]);
                reinterpret_cast<uint4*>(t->bytes)[v] = o;
            }
            if (tid < plan_words) t->plan[tid] = plan_stored[tid];  // (this thread's own store of a moment ago)
            if (tid == 0) {
                t->nbytes = sh.payload_bytes;
                t->plan_words = (uint32_t)plan_words;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(&t->state, tmpl_key, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // publish: the slot was written with write-through (sc1) stores; every storing wave drains them, the workgroup
    // meets, then one lane announces the slot (no release fence needed for sc1 payload: Guideline 16, R1).  A persistent
    // workgroup does not wait here: it announces the slot behind the barrier that ends the staging of its next slot
    // (publish_pending), when the stores have long drained.
    if (defer) {
        if (tid == 0) {
            defer->idx = idx;
            defer->done = done ? 1u : 0u;
            defer->silent = LACX_HOOK(prm, 8192u) ? 1u : 0u;
        }
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (done) fa.emitted[idx] = 2u;  // for the kernels that run after this one (k_pack)
        // test hook (bit 13): the slot is filled but never announced, so the packer gives up and k_pack takes over
        if (!LACX_HOOK(prm, 8192u)) rec_store(&fa.ready_rec[idx], done ? 1ull : 2ull);
    }
}

// The announcement of a slot whose stores every wave of the workgroup has drained (s_waitcnt vmcnt(0)) before the barrier
// the caller has just passed.
__device__ __forceinline__ void publish_pending(const FuseArgs& fa, PendingSlot& pend, int tid) {
    if (tid == 0) {
        const long long idx = pend.idx;
        if (idx >= 0) {
            if (pend.done) fa.emitted[idx] = 2u;
            if (!pend.silent) rec_store(&fa.ready_rec[idx], pend.done ? 1ull : 2ull);
        }
        pend.idx = -1;
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
                                                     bool ksums, uint32_t kmask) {
    uint64_t key = ~0ull;
    if (ksums) {  // planeTot[k] = sum_j (u_j >> k) already (ksums_wave), for the k of kmask (the others cannot win)
        if (lane < 16 && ((kmask >> lane) & 1u)) key = (((uint64_t)planeTot[lane] + (uint64_t)n * (uint64_t)(1 + lane)) << 4) | (uint64_t)lane;
    } else {
        // (planes below the lowest k of kmask were not counted: the suffix sums T_k of the k in the mask do not need them)
        const int b = 29 - lane;
        const uint64_t w = (lane < 30) ? ((uint64_t)planeTot[b] << b) : 0ull;
        const uint64_t tk = wave_scan_add_u64(w);  // lane l: T_(29-l)
        if (lane >= 14 && lane < 30 && ((kmask >> b) & 1u)) key = (((tk >> b) + (uint64_t)n * (uint64_t)(1 + b)) << 4) | (uint64_t)b;  // cost < 2^45
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
// Barrier between the phases of a slot.  A probe slot is one wave: its LDS accesses execute in order, so all it needs is
// that they have completed (and that the compiler keeps its order) -- no s_barrier, which lets the twelve probe slots of a
// block share a workgroup without sharing trip counts.
template <class G>
__device__ __forceinline__ void slot_sync() {
    if constexpr (G::T == 64) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
        __syncthreads();
    }
}

// The analysis of one slot (everything after the slot has been picked).
template <class G>
__device__ __forceinline__ void analyze_slot(unsigned char* smem_raw, const AnalyzeParams& prm, uint32_t n_in,
                                             const SlotSrc& src, int64_t start, const LpcSet* __restrict__ lpc_slot,
                                             ChannelPlan* __restrict__ plan_out, int tid, const FuseArgs& fuse,
                                             const long long fuse_idx, const bool fuse_flag_byte,
                                             const uint32_t fuse_flag_value, PendingSlot* pend) {
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
    const uint32_t chunk_bits = stage_samples(th, sh, src, start);
    // (digital silence: every wave says whether it saw a non-zero sample; wtotF is free until the first scan)
    {
        const bool wave_silent = __ballot(chunk_bits != 0u) == 0ull;
        if ((tid & 63) == 0) sh.wtotF[tid >> 6] = wave_silent ? 0u : 1u;
    }
    for (int i = tid; i < (int)(sizeof(LpcSet) / 2); i += G::T)
        reinterpret_cast<uint16_t*>(&sh.lpc)[i] = reinterpret_cast<const uint16_t*>(lpc_slot)[i];
    if (tid < 32) {
        sh.planeTot[0][tid] = sh.planeTot[1][tid] = 0;
        sh.planeTot256[0][tid] = sh.planeTot256[1][tid] = 0;
    }
    if (tid < 4) sh.acc[0][tid] = sh.acc[1][tid] = 0;
    if (tid < 33) (&sh.lbacc[0][0])[tid] = 0;
    if (tid < 2) sh.has4[tid] = 0;
    if (tid == 0) {
        sh.best_cand = -1;
        sh.tabUZ[G::T] = sh.tabUZ[G::T + 1] = 0;  // "not a zero" past the slot (phase_b_quick)
    }
    if (pend) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the previous slot's stores (long drained; free when there are none)
    slot_sync<G>();
    if (pend) publish_pending(fuse, *pend, tid);
    STAMP(0);

    // ---- pass 1: the pruning bound of every candidate ------------------------------------------------------------
    // One walk over the chunk for all eleven candidates (pass1_bounds, analyze_core.h): window loaded once, nothing
    // stored, no barrier between candidates (ref block/encoder.cpp:362-407 walks them one by one).  Per candidate the
    // thread holds the sum of its leading-bit counts and a 2-bit code per position (zero / four / other) that is counted
    // once per chunk.
    auto run_pass1 = [&](const bool real) {
        // Optimisation barrier on the chunk origin: without it the compiler hoists a dozen loop-invariant LDS
        // addresses and masks derived from it and, at the 128-VGPR budget, spills them to scratch.
        asm volatile("" : "+v"(th.a));
        const bool lpc_off = LACX_HOOK(prm, 16u);
        // bit_width(u | 1) + 1 = 33 - clz(u | 1) = 34 - lead_m per position.  A zero counts 2 that way and is worth 1 (the
        // wave's zero count takes the difference out); a position beyond the slot (r = 0) counts 2, is among those zeros
        // and is worth nothing (the wave's `beyond` takes the rest out).
        // Two candidates share a register for the wave sums (each sum stays below 2^16: 64 lanes x 34 x CH).
        const uint32_t per_thread = 34u * (uint32_t)G::CH;
        const int32_t left = (int32_t)n - (int32_t)((tid >> 6) * 64 * G::CH);  // samples of the slot from this wave's first one on
        const uint32_t valid = left <= 0 ? 0u : (left >= 64 * G::CH ? (uint32_t)(64 * G::CH) : (uint32_t)left);
        const uint32_t beyond = (uint32_t)(64 * G::CH) - valid;
        auto reduce_pair = [&](int c0, const BoundPartials& b0, bool two, const BoundPartials& b1) {
            const uint32_t lo = per_thread - b0.msum, hi = two ? per_thread - b1.msum : 0u;
            const uint32_t g2 = wave_sum_u32(lo | (hi << 16));
            const uint32_t cnt0 = wave_sum_u32(bound_counts<G::CH>(b0));
            const uint32_t cnt1 = two ? wave_sum_u32(bound_counts<G::CH>(b1)) : 0u;
            if ((tid & 63) == 0 && real) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 0 || two) {
                        const uint32_t g = h ? (g2 >> 16) : (g2 & 0xFFFFu);
                        const uint32_t cnt = h ? cnt1 : cnt0;
                        const uint32_t nz = cnt & 0x7FFu, n4 = (cnt >> 11) & 0x7FFu, ends = cnt >> 22;
                        atomicAdd(&sh.lbacc[c0 + h][0], g - nz - beyond);
                        atomicAdd(&sh.lbacc[c0 + h][1], (nz - beyond) + (n4 << 16));
                        atomicAdd(&sh.lbacc[c0 + h][2], ends);
                    }
                }
            }
        };
        // The fixed orders and the FIR predictor first, summed over the wave before the LPC candidates start: their six
        // sets of partials do not have to live through the LPC walks (where they did not fit the register file).
        {
            BoundPartials bf[6];
            if (n == (uint32_t)G::MAXN) pass1_bounds_fixed<G, true>(th, sh, bf);  // uniform
            else pass1_bounds_fixed<G, false>(th, sh, bf);
#pragma unroll
            for (int c = 0; c < 6; c += 2) reduce_pair(c, bf[c], true, bf[c + 1]);
        }
        asm volatile("" : "+v"(th.a));
        {
            BoundPartials bl[5];
            if (n == (uint32_t)G::MAXN) pass1_bounds_lpc<G, true>(th, sh, lpc_off, bl);  // uniform
            else pass1_bounds_lpc<G, false>(th, sh, lpc_off, bl);
            reduce_pair(6, bl[0], true, bl[1]);
            reduce_pair(8, bl[2], true, bl[3]);
            reduce_pair(10, bl[4], false, bl[4]);
        }
    };
    // A slot of nothing but zeros (digital silence) needs no bounds: every candidate's residual is the same zeros, so every
    // cost is the same and the lowest index wins (ref block/encoder.cpp:352-359 keeps the first strictly smallest cost):
    // candidate 0 is costed exactly, nothing else is looked at.
    bool silent = true;
#pragma unroll
    for (int w = 0; w < G::T / 64; ++w) silent = silent && sh.wtotF[w] == 0u;  // (uniform)
    uint32_t tmpl_key = 0;  // != 0: this slot is silent and its finished channel block is to be left in fuse.silent
    if (silent) {
        // Every silent slot of this length has the same plan and the same bitstream (under the same settings): where an
        // earlier one has left them behind, this one copies them.
        uint32_t key = 0;
        if constexpr (G::T == 1024) {
            if (fuse.silent && !LACX_HOOK(prm, ~0u)) {  // uniform
                key = 0x80000000u | (prm.zero_run ? 0x40000000u : 0u) | (prm.partitioning ? 0x20000000u : 0u) | n;
                if (tid == 0) sh.tmpl_state = __hip_atomic_load(&fuse.silent->state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        STAMP(2);
        slot_sync<G>();  // (every wave has read the flags before the scans reuse wtotF)
        if constexpr (G::T == 1024) {
            if (key) {  // uniform
                const uint32_t state = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.tmpl_state);
                if (state == key) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    const SilentTemplate* t = fuse.silent;
                    const uint32_t nbytes = t->nbytes;
                    if (tid == 0 && fuse.silent_copies) atomicAdd(fuse.silent_copies, 1u);
                    if ((uint32_t)tid < t->plan_words) reinterpret_cast<uint32_t*>(plan_out)[tid] = t->plan[tid];
                    if (fuse_idx >= 0) {  // uniform
                        if (tid == 0)
                            rec_store(&fuse.size_rec[fuse_idx], kRecValid | (fuse_flag_value ? kRecMs : 0ull) | (fuse_flag_byte ? kRecFlag : 0ull) |
                                                                    ((unsigned long long)nbytes + (fuse_flag_byte ? 1u : 0u)));
                        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                        uint8_t* slot = fuse.slots + (unsigned long long)fuse_idx * fuse.slot_stride;
                        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(slot, 0, (int)fuse.slot_stride, 0x00020000);
                        const uint32_t nv = (nbytes + 15u) >> 4;
                        for (uint32_t v = tid; v < nv; v += G::T) {
                            const uint4 w = reinterpret_cast<const uint4*>(t->bytes)[v];
                            u32x4 o;
                            o.x = w.x;
                            o.y = w.y;
                            o.z = w.z;
                            o.w = w.w;
                            __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, (int)(16u * v), 0, 16 /* sc1 */);
                        }
                        if (pend) {  // announced behind the next slot's staging barrier, like any other slot
                            if (tid == 0) {
                                pend->idx = fuse_idx;
                                pend->done = 1u;
                                pend->silent = 0u;
                            }
                        } else {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            __syncthreads();
                            if (tid == 0) {
                                fuse.emitted[fuse_idx] = 2u;
                                rec_store(&fuse.ready_rec[fuse_idx], 1ull);
                            }
                        }
                    }
                    return;
                }
                if (state == 0u && fuse_idx >= 0) tmpl_key = key;
            }
        }
        if (tid <= 10) sh.cand_key[tid] = tid == 0 ? 0ull : ~0ull;
    } else {
#if defined(LACX_STAMPS) && LACX_STAMPS == 3
    {   // diagnostic: the same pass (the same code) twice in a row -- the second trip finds it in the instruction cache
        uint32_t trips = 2;
        asm volatile("" : "+s"(trips));
        for (uint32_t r = 0; r < trips; ++r) {
            run_pass1(r == 0);
            if (r == 0) STAMP(2); else STAMP(15);
        }
    }
#else
    run_pass1(true);
    STAMP(2);
#endif
    slot_sync<G>();
    if (tid <= 10) {
        // one lane per candidate: its bound as a sortable key (bound * 16 + index), all ones when it is not available
        const bool avail = !((tid >= 6 && (sh.lpc.used[tid >= 6 ? tid - 6 : 0] == 0 || LACX_HOOK(prm, 16u))) ||
                             (tid >= 1 && LACX_HOOK(prm, 64u)));
        sh.cand_key[tid] = avail ? ((candidate_lower_bound(sh.lbacc[tid][0], sh.lbacc[tid][1], sh.lbacc[tid][2], n, prm.zero_run) << 4) | (uint64_t)tid)
                                 : ~0ull;
    }
    }  // (not silent)
    STAMP(5);

    // ---- pass 2: exact costs, most promising candidate first ----------------------------------------------------------
    // The reference keeps the first candidate with the strictly smallest cost = the minimum of (cost, index).  Candidates
    // are evaluated in ascending (bound, index) order; one that cannot beat the best (cost, index) so far ends the search,
    // because every remaining one has a bound at least as large (and, at an equal bound, a larger index).  A dismissed
    // candidate costs nothing here: no residual, no barrier.
    int pending = -1;        // candidate whose totals still have to be scored
    uint32_t pending_k0 = 0;
    bool pending_ksums = false;  // its totals are k-sums (32-bit blocks), not plane counts
    uint32_t pending_kmask = 0xFFFFu;  // ... for these k only
    uint32_t tried = 0;      // candidates already evaluated (uniform)
    int parity = 0;
    for (;;) {
        // (an opaque thread index per trip and again behind the loop: comparisons and addresses derived from it are loop
        // invariant, so the compiler computes them all in front of the loop -- two dozen lane masks and a handful of
        // addresses -- and, with nowhere to keep them, parks them in spill lanes and scratch memory across the search)
        asm volatile("" : "+v"(tid));
        if (tid < 64) {  // wave 0
            if (pending >= 0) {
                // previous candidate's totals sit in the other buffers: score it, then clear them
                score_candidate_wave(sh, pending, n, prm.zero_run, pending_k0, sh.planeTot[parity ^ 1], sh.acc[parity ^ 1], tid, pending_ksums, pending_kmask);
                if (tid < 32) sh.planeTot[parity ^ 1][tid] = sh.planeTot256[parity ^ 1][tid] = 0;
                if (tid < 4) sh.acc[parity ^ 1][tid] = 0;
            }
            // next: the untried candidate with the smallest (bound, index), unless it cannot win any more
            const uint64_t key = (tid <= 10 && !((tried >> tid) & 1u)) ? sh.cand_key[tid] : ~0ull;
            const uint64_t best_key = wave_last_u64(wave_scan_min_u64(key));
            if (tid == 0) {
                sh.has4[parity] = 0;
                sh.bqcount = 0;
                int next = best_key == ~0ull ? -1 : (int)(best_key & 15u);
                if (next >= 0 && !LACX_HOOK(prm, 128u) && candidate_pruned(best_key >> 4, next, sh.best_bits, sh.best_cand)) next = -1;
                sh.next_cand = next;
            }
        }
        STAMP(1);
        slot_sync<G>();  // Bsel: the previous candidate is scored, the next one chosen
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
        slot_sync<G>();  // B1b: the wave totals of the scan
        const uint64_t total_u = scan_pz_part2(sh, tid, sr);
        const bool narrow = total_u < kNarrowLimit;  // all prefix sums fit 32 bits (uniform)
        const bool ksums = narrow && !LACX_HOOK(prm, 262144u);
        // (the k whose static cost can still be the smallest follow from the block's sum alone: about four of sixteen)
        const uint32_t kmask = !LACX_HOOK(prm, 1048576u) ? static_k_candidates(total_u, n, tid & 63) : 0xFFFFu;
        if (!LACX_HOOK(prm, 1u)) {
            if (ksums) ksums_wave(th, pt, pt256, tid, kmask);
            else plane_totals_wave(th, pt, pt256, tid, kmask ? (int)__builtin_ctz(kmask) : 0);
        }
        // the first 256 samples all belong to wave 0: its own totals are complete once its atomics are (same wave,
        // program order), so it can derive the initial k at once; every other thread reads it after B3
        if (tid < 64) {
            const uint32_t k0w = initial_k_wave(pt256, n, tid, ksums);
            if (tid == 0) sh.cur_k0 = k0w;
        }
        STAMP(6);
        if (LACX_HOOK(prm, 2u)) {
            sh.tabF[tid] = 0;
            th.has4 = 1u;
        } else if (narrow) {
            phase_a<G, true>(th, sh);
        } else {
            phase_a<G, false>(th, sh);
        }
        if (__ballot(th.has4 != 0u) != 0ull && (tid & 63) == 0) sh.has4[parity] = 1u;  // read after B3
        STAMP(8);
        slot_sync<G>();  // B3: every chunk's flag counts are in tabF (phase B sums the six before its own), prefixes in tabP
        STAMP(10);
        const uint32_t k0 = sh.cur_k0;
        // (nothing derived from the thread index is carried from outside the candidate loop into phase B: the compiler
        // would keep a dozen LDS addresses alive across pass 1 and the loop, in scratch memory)
        asm volatile("" : "+v"(th.tid));
        if (LACX_HOOK(prm, 4u)) {
            th.crice = th.cbin = th.czr = 1;
            th.chasrun = 0;
        } else {
            // the zero-run cost only matters when the residual has a run of >= 4 zeros somewhere
            const bool zr = prm.zero_run && sh.has4[parity] != 0u;
            const bool full = n == (uint32_t)G::MAXN;
            if (G::T == 64 || LACX_HOOK(prm, 524288u)) {
                phase_b_dispatch<G>(th, sh, k0, narrow, zr, full);
                if ((uint32_t)th.a >= n) {
                    th.crice = th.cbin = th.czr = 0;
                    th.chasrun = 0;
                }
            } else {
                // Three chunks in four need no walk (phase_b_quick); the others are queued and walked after the barrier,
                // densely packed over the lanes of as few waves as it takes.
                const bool live = (uint32_t)th.a < n;
                bool quick = false;
                if (tid >= 64 && live) quick = phase_b_quick_dispatch<G>(th, sh, narrow, zr);
                if (!quick) {
                    th.crice = th.cbin = th.czr = 0;
                    th.chasrun = 0;
                }
                if (tid >= 64) {
                    const unsigned long long slow = __ballot(live && !quick);  // (wave-uniform)
                    if (slow != 0ull) {
                        uint32_t base = 0;
                        if ((tid & 63) == 0) base = atomicAdd(&sh.bqcount, (uint32_t)__popcll(slow));
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        if (live && !quick) sh.bqueue[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(slow >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)slow, 0u))] = (uint16_t)tid;
                    }
                }
                STAMP(11);
                slot_sync<G>();  // Bq: the queue is complete
                // Wave 0 walks its own chunks (the first 1024 samples, where the windows are still filling); waves 1..15
                // take the queue in blocks of 64, those that share a SIMD with wave 0 (4, 8, 12) last.
                if (tid < 64) {
                    if (live) phase_b_dispatch<G>(th, sh, k0, narrow, zr, full, false);
                } else {
                    const int w = tid >> 6;
                    const int order = (w & 3) ? (w - 1 - (w >> 2)) : (11 + (w >> 2));  // 1,2,3,5,6,7,... -> 0..11; 4,8,12 -> 12,13,14
                    const uint32_t qn = sh.bqcount;
                    for (uint32_t e0 = (uint32_t)order * 64u; e0 < qn; e0 += 15u * 64u) {  // wave-uniform trip count
                        const uint32_t e = e0 + (uint32_t)(tid & 63);
                        if (e < qn) phase_b_queued<G>(th, sh, (int)sh.bqueue[e], k0, narrow, zr, full);
                    }
                }
            }
        }
        STAMP(12);
        {
            const uint64_t r0 = wave_sum_u64(th.crice);
            const uint64_t r1 = wave_sum_u64(th.cbin);
            const uint64_t r2 = wave_sum_u64(th.czr);
            const uint32_t r3 = wave_or_u32(th.chasrun);
            if ((tid & 63) == 0) {
                atomicAdd(&acc[0], (unsigned long long)r0);
                atomicAdd(&acc[1], (unsigned long long)r1);
                atomicAdd(&acc[2], (unsigned long long)r2);
                atomicAdd(&acc[3], (unsigned long long)r3);
            }
        }
        STAMP(13);
        slot_sync<G>();  // B5
        STAMP(14);
        pending = cand;
        pending_k0 = k0;
        pending_ksums = ksums;
        pending_kmask = kmask;
        parity ^= 1;
    }

    STAMP(15);
    asm volatile("" : "+v"(tid), "+v"(th.tid), "+v"(th.a));
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
    if (best == pending && !LACX_HOOK(prm, 16384u)) {
        // The winner is the candidate evaluated last (usually the only one): its residual is still in sh.u (plain: the
        // micro-window flags of phase A live in their own tables), its prefix sums in tabP / tabNZ, its plane counts in
        // th.cs.  Nobody reads the staged samples any more (the last barrier of the search is behind us).
        clear_partition_scratch();
        slot_sync<G>();
    } else {
        phase_r(th, sh, best);  // last reader of the staged samples; leaves the plain residual in sh.u
        ScanRegs<G> sr;
        scan_pz_part1(sh, tid, sr);
        slot_sync<G>();
        clear_partition_scratch();
        scan_pz_part2(sh, tid, sr);
        slot_sync<G>();
    }
    const bool pnarrow = sh.tabP[G::T] < kNarrowLimit;
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
        slot_sync<G>();
        {
            constexpr int NW = G::T / 64;
            const int wave = tid >> 6, lane = tid & 63;
            for (int w = wave; w < 15; w += NW) wave_exclusive_scan_u32(pm.grp[w], G::NG + 1, lane);
        }
        slot_sync<G>();
        STAMP(17);
        for (int idx = tid; idx < (LACX_HOOK(prm, 32u) ? 0 : nseg); idx += G::T) {
            const int p = 31 - __clz(idx + 2);
            seg_static_eval(sh, n, p, (uint32_t)(idx + 2 - (1 << p)));
        }
        slot_sync<G>();
        STAMP(18);
        auto flush = [&pm](uint32_t idx, unsigned long long rc, unsigned long long bn, unsigned long long zr,
                           uint32_t hr) {
            atomicAdd(&pm.segacc[idx][0], rc);
            atomicAdd(&pm.segacc[idx][1], bn);
            atomicAdd(&pm.segacc[idx][2], zr);
            if (hr) atomicOr(&pm.segrun[idx], 1u);
        };
        if (LACX_HOOK(prm, 8u)) {
            // (timing ablation only)
        } else if (pnarrow && partitions_chunk_aligned<G>(n, max_p)) {
            // all orders in one walk (every full block, every probe); without a run of >= 4 zeros in the
            // block no partition can have one, so the zero-run costs are not needed.  Narrow sums: every segment total
            // stays below 2^32 (sum of u < kNarrowLimit, at most 36 bits of overhead per sample), so 32-bit LDS atomics on the
            // low words of the (zeroed) 64-bit accumulators suffice.
            // One atomic per accumulator and segment of neighbouring lanes (see seg_sum_u32): the lanes of a partition of
            // order p are (n >> p) / CH neighbours -- a power of two for every full block and every probe; other sizes
            // fall back to one atomic per lane.  Called by every lane of the wave (idle lanes pass zeros).
            const bool ablate_flush = LACX_HOOK(prm, 65536u);
            const uint32_t chunks = n / (uint32_t)G::CH;  // chunks of the slot
            const bool pow2 = (chunks & (chunks - 1u)) == 0u && !LACX_HOOK(prm, 131072u);
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
            auto flush_entry = [&pm, ablate_flush](uint32_t idx, uint32_t rc, uint32_t bn, uint32_t zr, uint32_t hr) {
                if (ablate_flush) return;
                atomicAdd(reinterpret_cast<uint32_t*>(&pm.segacc[idx][0]), rc);
                atomicAdd(reinterpret_cast<uint32_t*>(&pm.segacc[idx][1]), bn);
                if (zr) atomicAdd(reinterpret_cast<uint32_t*>(&pm.segacc[idx][2]), zr);
                if (hr) atomicOr(&pm.segrun[idx], 1u);
            };
            const bool with_zr = prm.zero_run && sh.best_hasrun;  // (block-uniform)
            if (LACX_HOOK(prm, 32768u)) {  // (A/B: the plain walk over every sample and order)
                if (with_zr) partition_fused<G, true>(th, sh, max_p, flush32);
                else partition_fused<G, false>(th, sh, max_p, flush32_nozr);
            } else {
                // no sample walk where the Rice parameter is provably constant over the chunk; the other (chunk, order)
                // pairs are queued and walked densely packed
                // (per wave: no workgroup barrier, no atomic -- a wave's queue is filled and drained by the wave itself)
                uint16_t* wq = &pm.queue[(tid >> 6) * 64 * G::MAXP];
                uint32_t queued = 0;  // wave-uniform
                auto enqueue = [&](uint32_t entry, bool ambiguous) {
                    const unsigned long long m = __ballot(ambiguous);
                    if (ambiguous) wq[queued + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)entry;
                    queued += (uint32_t)__popcll(m);
                };
                asm volatile("" : "+v"(th.tid));  // (nothing derived from the thread index lives on from the phases before)
                if (!with_zr) {
                    partition_quick<G>(th, sh, max_p, flush32_nozr, enqueue);
                } else {
                    // With zero-run costs: where most of a wave's pairs would need the walk anyway (a residual whose mean
                    // sits on a parameter boundary) the walk over all orders at once is cheaper than classifying first:
                    // the wave decides for itself (any mix of the two gives the same sums).
                    QuickPrep<G> qp;
                    partition_quick_prepare<G>(th, sh, max_p, qp);
                    uint32_t pairs = 0;  // wave-uniform
                    for (int q = 0; q < max_p; ++q) pairs += (uint32_t)__popcll(__ballot((qp.amb >> q) & 1u));
                    if (pairs > kQuickMaxPairs) partition_fused<G, true>(th, sh, max_p, flush32);
                    else partition_quick_costs<G>(th, sh, max_p, qp, flush32, enqueue);
                }
                STAMP(7);
                // The queued pairs cluster where partitions begin (the prefix mean still moves there), so the waves' queues
                // differ severalfold in length and, walked wave by wave, the longest one kept the other fifteen waves
                // waiting at the barrier behind the search.  One barrier here instead: every wave publishes its count, and
                // the pairs of ALL queues are walked in equal shares -- entry e of the concatenated queues by thread e % T.
                if constexpr (G::T > 64) {
                    constexpr int NW = G::T / 64;
                    static_assert(NW <= 16, "one count per wave");
                    const int lane = tid & 63;
                    if (lane == 0) pm.wqcount[tid >> 6] = queued;
                    __syncthreads();
                    const uint32_t mine = lane < NW ? pm.wqcount[lane] : 0u;
                    const uint32_t incl = wave_scan_add_u32(mine);             // lanes 0..NW-1: entries up to and including wave `lane`
                    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, NW - 1);
                    for (uint32_t e0 = (uint32_t)(tid & ~63); e0 < total; e0 += (uint32_t)G::T) {  // wave-uniform trip count:
                        // every lane takes part in the shuffles (a disabled source lane would read as zero)
                        const bool live = e0 + (uint32_t)lane < total;
                        const uint32_t e = live ? e0 + (uint32_t)lane : total - 1u;
                        // the wave whose queue holds entry e: the first w with incl[w] > e (four shuffle steps over 16 lanes)
                        uint32_t w = 0;
#pragma unroll
                        for (int step = NW / 2; step >= 1; step >>= 1) {
                            const uint32_t probe = (uint32_t)__shfl((int)incl, (int)(w + (uint32_t)step - 1u), 64);
                            w += probe <= e ? (uint32_t)step : 0u;
                        }
                        // (unconditional: under a per-lane condition the source lanes that skip it would read as zero)
                        const uint32_t prev = (uint32_t)__shfl((int)incl, (int)(w ? w - 1u : 0u), 64);
                        const uint32_t before = w ? prev : 0u;
                        const uint32_t entry = pm.queue[w * 64u * (uint32_t)G::MAXP + (e - before)];
                        if (live) {
                            if (with_zr) partition_slow_entry<G, true>(sh, n, entry, flush_entry);
                            else partition_slow_entry<G, false>(sh, n, entry, flush_entry);
                        }
                    }
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS stores, before it reads them back
                    for (uint32_t e = (uint32_t)(tid & 63); e < queued; e += 64u) {
                        if (with_zr) partition_slow_entry<G, true>(sh, n, wq[e], flush_entry);
                        else partition_slow_entry<G, false>(sh, n, wq[e], flush_entry);
                    }
                }
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
        slot_sync<G>();
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
        slot_sync<G>();
    }
    if (tid == 0) finalize_plan(sh, n, prm.zero_run, max_p, &sh.plan);
    slot_sync<G>();
    // only the head and the partitions in use: the rest of the record is zero already (the plans are cleared per call)
    const int plan_words = (int)(offsetof(ChannelPlan, part_mode_k) + ((size_t)1 << sh.plan.partition_order) + 3) / 4;
    for (int i = tid; i < plan_words; i += G::T)
        reinterpret_cast<uint32_t*>(plan_out)[i] = reinterpret_cast<const uint32_t*>(&sh.plan)[i];
    STAMP(21);
    if constexpr (G::T == 1024) {
        if (fuse_idx >= 0) fused_emit<G>(sh, th, prm, fuse, fuse_idx, fuse_flag_byte, fuse_flag_value, tid, pend, tmpl_key,
                                        reinterpret_cast<const uint32_t*>(plan_out), plan_words STAMP_ARGS);  // uniform
    }
    STAMP(23);
#if defined(LACX_STAMPS) && (LACX_STAMPS == 1 || LACX_STAMPS == 3)
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
#ifndef LACX_PROBE_WAVES
#define LACX_PROBE_WAVES 6
#endif
// Probe class: one wave per probe slot, one slot per workgroup.  (Measured against the twelve probe slots of a block as
// the twelve waves of ONE workgroup -- independent, no workgroup barrier (slot_sync), but started together so that the
// instruction stream is fetched once for twelve waves: white noise, every block probed, 0.69 -> 0.65 ms; music, 28 % of
// the blocks probed, 0.18 -> 0.26 ms, because 492 twelve-wave workgroups balance worse over 256 CUs than 5 904 single
// waves.  LACX_PROBE_WG_WAVES=12 rebuilds that form.)
#ifndef LACX_PROBE_WG_WAVES
#define LACX_PROBE_WG_WAVES 1
#endif
constexpr int kProbeWaves = LACX_PROBE_WG_WAVES;
// What only the probe class is handed (front_ctr, kernels_internal.h) -- as kernel arguments of the whole-block class the
// three pointers cost it a spilled VGPR and sixteen spilled SGPRs although it never looks at them.
template <class G>
struct TailArgs {};
template <>
struct TailArgs<GProbe> {
    uint32_t* front_ctr;
    BlockPlan* bplans;
    uint32_t* need_full_out;
};
template <class G>
__global__ __launch_bounds__(G::T == 64 ? 64 * kProbeWaves : G::T, G::T == 64 ? LACX_PROBE_WAVES : 4) void k_analyze(BatchRef br, int probe_class, uint32_t one_block,
                                                  int which_base, const LpcSet* __restrict__ lpcs,
                                                  const uint32_t* __restrict__ need,
                                                  ChannelPlan* __restrict__ plans,
                                                  unsigned long long* __restrict__ t_first,
                                                  unsigned long long* __restrict__ t_last, FuseArgs fuse,
                                                  uint32_t* __restrict__ work_ctr, uint32_t total_wg, uint32_t pair_blocks,
                                                  TailArgs<G> tail) {
    extern __shared__ __align__(16) unsigned char smem_all[];
    __shared__ uint32_t s_next;
    __shared__ PendingSlot s_pend;
    static_assert(sizeof(Smem<G>) % 16 == 0, "slot images are 16-byte aligned");
    const int tid = G::T == 64 ? (int)(threadIdx.x & 63u) : (int)threadIdx.x;
    unsigned char* smem_raw = smem_all + (G::T == 64 ? (size_t)(threadIdx.x >> 6) * sizeof(Smem<G>) : 0);
    // the earliest start, kept inverted (the word starts as zero like everything else the call clears)
    if (t_first && tid == 0) atomicMax(t_first, ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
    // Persistent form (work_ctr != nullptr; whole-block class only): one workgroup per CU takes virtual workgroup ids from
    // a counter until none is left, instead of one launched workgroup per slot.  A 1024-thread workgroup that owns a
    // whole CU costs a turnaround when it retires (all sixteen waves gone, LDS released, sixteen new waves set up), and
    // its last act was waiting for its slot stores to drain before announcing the slot; the persistent workgroup goes
    // straight on to staging its next slot and announces the previous one behind that staging's barrier.  ONE counter
    // for the whole chip: with a counter per XCD (to keep the two slots of a block on one XCD, as the launched grid
    // does) the XCDs that lend a CU to the streaming packer finish 3 % later than the others, which costs more than the
    // sibling's L2 hit is worth.  Dynamic, so the workgroups that find no free CU while the packer holds its three simply
    // find no work left when they start.
    const bool persistent = work_ctr != nullptr;  // (uniform)
    // Work units of the persistent form over ONE stereo stream (pair_blocks > 0): unit v < pair_blocks is BLOCK v -- the
    // workgroup analyses its two channel slots one after the other, so the block's PCM comes from HBM once (the second
    // staging hits the L2 / TCP of the same CU) instead of once per XCD that happens to draw one of its slots; the last
    // blocks of the stream are handed out slot by slot (units pair_blocks ..), so that the tail of the kernel is balanced
    // in single slots as before.
    uint32_t rep = 0;         // channel slot of a pair unit
    uint32_t v = G::T == 64 ? blockIdx.x * (uint32_t)kProbeWaves + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : blockIdx.x;  // virtual workgroup id
    if (persistent) {
        if (tid == 0) {
            s_next = atomicAdd(work_ctr, 1u);
            s_pend.idx = -1;
        }
        __syncthreads();
        v = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_next);  // (uniform: into a scalar register, and with it everything derived from it)
    }
    for (;;) {
        if (persistent && v >= total_wg) break;
        // Dense grids: consecutive workgroups are dealt round-robin to the 8 XCDs, so every launched workgroup
        // should be one that has work.  Whole-block class: the stream's workgroup w analyses the (w % per)-th needed slot
        // of its block w / per (per = the stream's channels; the streams of a set follow each other in the grid).  Probe
        // class: 12 slots per block, skipped unless the block is uncertain.  which_base != 0: the two extra workgroups of
        // ONE block (global block one_block) whose four channels are all needed.
        uint32_t blk;  // global block of the launch set
        uint32_t reps = 1u;  // slots this unit consists of
        int slot = -1;
        int which_in_block = -1;   // position of the slot among the block's needed whole-block slots
        uint32_t needed_slots = 0;
        StreamDesc sd;
        if (probe_class) {
            blk = v / 12u;
            sd = stream_of_block_uniform(br, blk);
            const int s = 4 + (int)(v % 12u);
            if (((uint32_t)__builtin_amdgcn_readfirstlane((int)need[blk]) >> s) & 1u) slot = s;
        } else {
            uint32_t wsel;
            if (which_base) {
                blk = one_block;
                sd = stream_of_block_uniform(br, blk);
                wsel = v;
            } else if (persistent && pair_blocks != 0u) {  // (one stream: the descriptor is in the kernel arguments)
                sd = br.single;
                uint32_t lblk;
                if (v < pair_blocks) {
                    lblk = v;
                    wsel = rep;
                    reps = 2u;
                } else {
                    const uint32_t s = v - pair_blocks;
                    lblk = pair_blocks + (s >> 1);
                    wsel = s & 1u;
                }
                blk = sd.first_block + lblk;
            } else {
                sd = stream_of_workgroup(br, v);
                const uint32_t per = sd.prm.channels == 2 ? 2u : 1u;
                uint32_t lblk;
                xcd_slot(v - sd.first_wg, per, LACX_HOOK(sd.prm, 512u) ? 0u : sd.prm.num_blocks, lblk, wsel);
                blk = sd.first_block + lblk;
            }
            int which = (int)wsel + which_base;
            which_in_block = which;
            uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)need[blk]) & 0xFu;
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
        if (slot >= 0) {  // uniform for the workgroup
            const AnalyzeParams prm = sd.prm;
            const uint32_t lblk = blk - sd.first_block;
            const SlotGeom g = slot_geom(prm, lblk, slot);
            const uint32_t n = g.n;
            const size_t sidx = (size_t)blk * kSlotsPerBlock + slot;
            const SlotSrc src = slot_src(prm, sd.left, sd.right, slot & 3);

            // Fused emit: only where the block's channel pair is already final, i.e. exactly `channels` whole-block slots
            // are needed (a small final block that is encoded both ways and compared afterwards is left to k_emit).
            long long fuse_idx = -1;
            bool flag_byte = false;
            if (fuse.slots && !probe_class && !which_base) {
                const uint32_t item = lblk * (uint32_t)prm.channels + (uint32_t)which_in_block;  // within the stream
                fuse_idx = (long long)prm.stream_base + item;
                flag_byte = prm.channels == 2 && prm.stereo_mode == 2 && which_in_block == 0;
                // the host excludes a small final block that may be encoded both ways and compared afterwards (fuse_items)
                if (item >= sd.fuse_items || needed_slots != (uint32_t)prm.channels) fuse_idx = -1;
            }
            // (an opaque copy of the thread index per slot: otherwise everything the analysis derives from it is invariant
            // in the persistent loop, hoisted out of it and kept alive in scratch memory)
            int slot_tid = tid;
            asm volatile("" : "+v"(slot_tid));
            analyze_slot<G>(smem_raw, prm, n, src, g.start, &lpcs[sidx], &plans[sidx], slot_tid, fuse, fuse_idx, flag_byte,
                            (uint32_t)((slot & 3) >= 2 ? 1u : 0u), persistent ? &s_pend : nullptr);
            if (t_last && tid == 0) atomicMax(t_last, (unsigned long long)__builtin_amdgcn_s_memrealtime());
            if constexpr (G::T == 64) {
                // Probe class: the last of a block's twelve probe slots makes the block's LR/MS choice (front_ctr,
                // kernels_internal.h).  The slot's size goes out once more with an agent-scope store and is waited for
                // in front of the count; the last wave reads the twelve sizes with agent-scope loads (no fences).
                uint32_t* const front_ctr = tail.front_ctr;
                if (probe_class && front_ctr) {  // uniform
                    uint32_t last = 0;
                    if (tid == 0) {
                        ChannelPlan* po = &plans[sidx];
                        agent_store(&po->payload_bytes, reinterpret_cast<Smem<G>*>(smem_raw)->plan.payload_bytes);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        last = atomicAdd(&front_ctr[(size_t)blk * 2 + 1], 1u) == 11u ? 1u : 0u;
                    }
                    if (__builtin_amdgcn_readfirstlane((int)last)) {
                        if (tid == 0) front_ctr[(size_t)blk * 2 + 1] = 0u;
                        if (tid < 16) decide_probed_block(blk, tid, tail.bplans, tail.need_full_out, plans);
                    }
                }
            }
        }
        if (!persistent) break;
        __syncthreads();  // every wave is done with this slot's LDS image (the emit's tile aliases the next staging area)
        if (++rep < reps) continue;  // (uniform) the block's other channel slot
        rep = 0;
        if (tid == 0) s_next = atomicAdd(work_ctr, 1u);
        __syncthreads();
        v = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_next);
    }
    if (persistent) {  // the last slot of a persistent workgroup
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        publish_pending(fuse, s_pend, tid);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
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
hipError_t ensure_kernel_attrs() {
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
    if (e == hipSuccess) e = set_kernel_attrs_front();
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_analyze<GProbe>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(Smem<GProbe>) * kProbeWaves));
    if (e == hipSuccess) e = set_kernel_attrs_emit();
    if (e == hipSuccess) done[dev] = true;
    return e;
}

// CUs of the current device (cached per device ordinal).
static uint32_t compute_units() {
    static std::mutex mu;
    static int cus[kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256u;
    std::lock_guard<std::mutex> lock(mu);
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return (uint32_t)cus[dev];
}

hipError_t launch_analysis(const LaunchSet& ls, const DeviceWorkspace& ws, hipStream_t stream, hipEvent_t* ev,
                           const FuseArgs* fuse, hipEvent_t wait_before_full, const LaunchTuning& tune) {
    const FuseArgs fa = fuse ? *fuse : FuseArgs{};
    hipError_t e = ensure_kernel_attrs();
    if (e != hipSuccess) return e;
    const BatchRef& br = ls.br;
    const uint32_t nb = br.total_blocks;
    if (nb == 0) return hipSuccess;
    hipStream_t full_stream = stream;  // the whole-block kernel and everything behind it
    if (tune.front_stream && ev) stream = tune.front_stream;  // the front kernels (needs ev[2] to order the two)
    if (ev) (void)hipEventRecord(ev[0], stream);
    e = hipMemsetAsync(ws.plans, 0, sizeof(ChannelPlan) * (size_t)nb * kSlotsPerBlock, stream);
    if (e != hipSuccess) return e;
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
    // the front kernels of the blocks [b0, b0 + cnt) of a one-stream set: ingest, stereo estimate, Levinson, probes, decision
    auto launch_front = [&](const BatchRef& r, const DeviceWorkspace& w, uint32_t cnt, hipStream_t st, bool mark) {
        uint32_t* fc = tune.fold_front ? w.front_ctr : nullptr;
        hipLaunchKernelGGL(k_ingest, dim3(cnt * 4u), dim3(kIngestThreads), 0, st, r, w.sums, w.badidx, w.acorr, fc, w.bplans,
                           w.need_probe, w.need_full);
        if (!fc) hipLaunchKernelGGL(k_stereo, dim3((cnt + 3) / 4), dim3(64), 0, st, r, w.sums, w.badidx, w.bplans, w.need_probe, w.need_full);
        hipLaunchKernelGGL(k_levinson, dim3((cnt * kSlotsPerBlock + kLevThreads - 1) / kLevThreads), dim3(kLevThreads),
                           sizeof(LevMem), st, r, w.acorr, w.need_probe, w.lpcs);
        if (mark && ev) (void)hipEventRecord(ev[1], st);
        if (any_auto) {
            static_assert(12 % kProbeWaves == 0, "whole workgroups per block");
            hipLaunchKernelGGL(k_analyze<GProbe>, dim3(cnt * (12u / kProbeWaves)), dim3(GProbe::T * kProbeWaves), sizeof(Smem<GProbe>) * kProbeWaves, st, r, 1, 0u, 0,
                               w.lpcs, w.need_probe, w.plans, (unsigned long long*)nullptr, (unsigned long long*)nullptr, FuseArgs{},
                               (uint32_t*)nullptr, 0u, 0u, TailArgs<GProbe>{fc, w.bplans, w.need_full});
            if (!fc) hipLaunchKernelGGL(k_decide, dim3((cnt + 3) / 4), dim3(64), 0, st, r, 1, w.bplans, w.need_probe, w.need_full, w.plans);
        }
    };
    const bool halves = tune.aux_stream && tune.aux_ev[0] && tune.aux_ev[1] && br.table == nullptr && ls.nstreams == 1 && nb >= 1024u;
    if (halves) {
        const AnalyzeParams& p = br.single.prm;
        const uint32_t nb1 = (nb / 2u) & ~7u;  // (whole groups of eight blocks: the XCD-aware numbering of k_ingest)
        const uint64_t frame_bytes = p.layout == PCM_INTERLEAVED_I16 ? 2ull * (uint64_t)p.channels
                                                                     : (p.layout == PCM_INTERLEAVED_I24 ? 3ull * (uint64_t)p.channels : 4ull);
        auto sub = [&](uint32_t b0, uint32_t cnt) {
            BatchRef r = br;
            StreamDesc& sd = r.single;
            const uint64_t f0 = (uint64_t)b0 * kMaxBlock;
            const uint64_t f1 = p.frames < (uint64_t)(b0 + cnt) * kMaxBlock ? p.frames : (uint64_t)(b0 + cnt) * kMaxBlock;
            sd.prm.frames = f1 - f0;
            sd.prm.num_blocks = cnt;
            r.total_blocks = cnt;
            sd.left = reinterpret_cast<const int32_t*>(reinterpret_cast<const uint8_t*>(br.single.left) + f0 * frame_bytes);
            sd.right = (p.layout == PCM_PLANAR_I32 && br.single.right) ? br.single.right + f0 : nullptr;
            return r;
        };
        auto wsub = [&](uint32_t b0) {
            DeviceWorkspace w = ws;
            const size_t sl = (size_t)b0 * kSlotsPerBlock;
            w.plans += sl;
            w.bplans += b0;
            w.need_probe += b0;
            w.need_full += b0;
            w.acorr += sl * 13;
            w.lpcs += sl;
            w.sums += (size_t)b0 * 12;
            w.badidx += (size_t)b0 * 2;
            if (w.front_ctr) w.front_ctr += (size_t)b0 * 2;
            return w;
        };
        hipError_t he = hipEventRecord(tune.aux_ev[0], stream);  // behind the memset and whatever the caller queued before
        if (he == hipSuccess) he = hipStreamWaitEvent(tune.aux_stream, tune.aux_ev[0], 0);
        if (he != hipSuccess) return he;
        launch_front(sub(0, nb1), ws, nb1, stream, true);
        launch_front(sub(nb1, nb - nb1), wsub(nb1), nb - nb1, tune.aux_stream, false);
        he = hipEventRecord(tune.aux_ev[1], tune.aux_stream);
        if (he == hipSuccess) he = hipStreamWaitEvent(stream, tune.aux_ev[1], 0);
        if (he != hipSuccess) return he;
    } else {
        launch_front(br, ws, nb, stream, true);
    }
    if (ev) (void)hipEventRecord(ev[2], stream);
    if (full_stream != stream) {
        const hipError_t we = hipStreamWaitEvent(full_stream, ev[2], 0);
        if (we != hipSuccess) return we;
        stream = full_stream;
    }
    if (wait_before_full) {
        const hipError_t we = hipStreamWaitEvent(stream, wait_before_full, 0);
        if (we != hipSuccess) return we;
    }
    // Persistent form where the caller provides the (zeroed) work counters: one workgroup per CU.  The streaming packer
    // is resident by then (it is started in front of the ingest kernel in this form); the two or three workgroups that
    // find its CUs taken wait, start when the others have left, find no work and leave.
    const bool persistent = analysis_is_persistent(ws);
    const uint32_t pgrid = tune.persistent_grid ? tune.persistent_grid : compute_units();
    // pair units (see k_analyze): one stereo stream, all blocks but the last `pgrid` (those go out slot by slot)
    uint32_t pair_blocks = 0, units = total_wg;
    if (persistent && !tune.no_pairs && br.table == nullptr && ls.nstreams == 1 && ls.streams[0].prm.channels == 2 && nb > pgrid) {
        pair_blocks = nb - pgrid;
        units = pair_blocks + 2u * (nb - pair_blocks);
    }
    const uint32_t grid = persistent ? (units < pgrid ? units : pgrid) : total_wg;
    hipLaunchKernelGGL(k_analyze<GFull>, dim3(grid), dim3(GFull::T), sizeof(Smem<GFull>), stream, br, 0, 0u, 0,
                       ws.lpcs, ws.need_full, ws.plans, ws.t_first, ws.t_last, fa, persistent ? ws.work_ctr : (uint32_t*)nullptr,
                       units, pair_blocks, TailArgs<GFull>{});
    if (any_both) {  // the 3rd and 4th slots of such a final block: a two-workgroup launch each
        for (uint32_t i = 0; i < ls.nstreams; ++i) {
            const StreamDesc& sd = ls.streams[i];
            const AnalyzeParams& p = sd.prm;
            if (p.channels == 2 && p.stereo_mode == 2 && p.frames - (uint64_t)(p.num_blocks - 1) * kMaxBlock <= (uint64_t)kFullCompareLimit)
                hipLaunchKernelGGL(k_analyze<GFull>, dim3(2), dim3(GFull::T), sizeof(Smem<GFull>), stream, br, 0,
                                   sd.first_block + p.num_blocks - 1u, 2, ws.lpcs, ws.need_full, ws.plans,
                                   (unsigned long long*)nullptr, (unsigned long long*)nullptr, FuseArgs{}, (uint32_t*)nullptr, 0u, 0u,
                                   TailArgs<GFull>{});
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
