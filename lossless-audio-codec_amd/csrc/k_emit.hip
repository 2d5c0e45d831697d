// k_emit.hip -- everything behind the analysis: offsets, packing, repair emit, result gather.
//   k_offsets   one workgroup: block byte sizes -> payload offsets + the container's block table
//   k_stream_out        beside the whole-block analysis, on its own stream: packer waves move the staging slots of the
//                       fused emit to the payload (device memory drained by a copy engine, or pinned host memory)
//   k_pack, k_emit<16,1024>   repair paths: slots the packer did not move / channel blocks the fused emit left out
//                       (k_emit alone is the whole emit with LACX_FUSED_EMIT=0)
//   k_gather    block plans, block table, totals and flags into pinned host memory
// The host emit (emit.cpp, LACX_FLAG_HOST_EMIT) consumes the same ChannelPlan records instead of k_offsets/k_emit.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "emit_device.h"
#include "kernels_internal.h"

namespace lacx {

// ---------------------------------------------------------------------------------------------
// device-side emit: k_offsets + k_emit
// ---------------------------------------------------------------------------------------------
// One workgroup: byte size of every block's payload ([flag] + the two chosen channel blocks), exclusive
// prefix -> block_off[0..nb] (byte offset of the block in the result buffer), and the container's block table entries
// (frames, bytes).  In a set of several streams every stream's payload starts at its own region (StreamDesc::out_base):
// stream_pre[s] receives the prefix at the stream's first block and the offsets are re-based per stream.
#ifdef LACX_STAMPS
__device__ unsigned long long g_off_stamps[8];  // (diagnostic: 100 MHz realtime stamps of thread 0 of k_offsets)
#define OFF_STAMP(i) do { if (threadIdx.x == 0) g_off_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define OFF_STAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(1024) void k_offsets(BatchRef br, const BlockPlan* __restrict__ bplans,
                                                   const ChannelPlan* __restrict__ plans,
                                                   unsigned long long* __restrict__ block_off,
                                                   uint32_t* __restrict__ table,
                                                   const unsigned long long* __restrict__ base_ptr,
                                                   unsigned long long* __restrict__ stream_pre) {
    __shared__ unsigned long long s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    OFF_STAMP(0);
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
    OFF_STAMP(1);  // (the sum depends on every load: they are back)
    const unsigned long long inc = wave_scan_add_u64(sum);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    OFF_STAMP(2);
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
    OFF_STAMP(3);
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
// host side
// ---------------------------------------------------------------------------------------------
hipError_t set_kernel_attrs_emit() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_emit<GFull>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)sizeof(EmitMem<GFull>));
}

int debug_read_offset_stamps(unsigned long long* out8) {  // (diagnostic builds only)
#ifdef LACX_STAMPS
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_off_stamps), sizeof(unsigned long long) * 8) == hipSuccess ? 1 : 0;
#else
    (void)out8;
    return 0;
#endif
}

hipError_t launch_emit(const LaunchSet& ls, const DeviceWorkspace& ws, uint8_t* out,
                       const unsigned long long* base_ptr, hipEvent_t wait_before_offsets,
                       hipEvent_t offsets_done, hipStream_t stream, bool skip_emitted, const uint32_t* moved_total,
                       uint32_t shard_items, hipEvent_t wait_before_pack, uint32_t* repacked, bool lazy_repair) {
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
    if (lazy_repair) return hipGetLastError();
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
                             hipStream_t stream, const RangeProgress& rp, const LaunchTuning& tune) {
    if (ls.total_items == 0) return hipSuccess;
    const int nap = tune.pack_nap > 0 ? tune.pack_nap : 1;
    const int grid = tune.pack_grid > 0 ? tune.pack_grid : (rp.host_end ? kStreamGridDevice : kStreamGrid);
    // counters: [0] error flags, [1] channel blocks put in place, [2] packer waves that gave up waiting
    hipLaunchKernelGGL(k_stream_out, dim3((uint32_t)grid), dim3(kStreamThreads), 0, stream, ls.br, ls.total_items, nap, ls.item_stream,
                       (const unsigned long long*)ws.size_rec, (const unsigned long long*)ws.ready_rec, (const uint8_t*)ws.slots,
                       ws.slot_stride, out, ws.packed, counters, counters + 1, counters + 2, rp);
    return hipGetLastError();
}

}  // namespace lacx
