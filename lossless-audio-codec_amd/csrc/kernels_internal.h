// kernels_internal.h -- what the kernel translation units share besides device_util.h: geometry aliases, kernel
// prototypes for the launchers, the per-unit attribute setters.
#pragma once
#include "device_util.h"
#include "emit_core.h"

namespace lacx {

using GFull = Geo<16, 1024>;
using GProbe = Geo<4, 64>;
constexpr int kMaxDevices = 64;

// k_front.hip
constexpr int kIngestThreads = 256;
constexpr int kLevThreads = 256;
struct LevMem {  // work arrays R, a, prevA of every thread, one column per thread (120 KiB)
    uint64_t m[3][13][kLevThreads];
    uint32_t es[3][13][kLevThreads];  // sign << 31 | (exponent + 2^21)
};
// front_ctr ([block][2], zero between calls; null: off): the last of a block's four ingest workgroups makes the block's
// stereo estimate itself (what k_stereo does), the last of its twelve probe slots the LR/MS choice (k_decide phase 1) --
// two kernels fewer in the dependent chain in front of the whole-block analysis.  Whoever finds the count complete puts
// the word back to zero.
__global__ void k_ingest(BatchRef br, unsigned long long* __restrict__ sums, uint32_t* __restrict__ badidx,
                         int64_t* __restrict__ acorr, uint32_t* __restrict__ front_ctr, BlockPlan* __restrict__ bplans,
                         uint32_t* __restrict__ need_probe, uint32_t* __restrict__ need_full);
__global__ void k_stereo(BatchRef br, const unsigned long long* __restrict__ sums, const uint32_t* __restrict__ badidx,
                         BlockPlan* __restrict__ bplans, uint32_t* __restrict__ need_probe, uint32_t* __restrict__ need_full);
__global__ void k_levinson(BatchRef br, const int64_t* __restrict__ acorr, const uint32_t* __restrict__ need_probe,
                           LpcSet* __restrict__ lpcs);
__global__ void k_decide(BatchRef br, int phase, BlockPlan* __restrict__ bplans, const uint32_t* __restrict__ need_probe,
                         uint32_t* __restrict__ need_full, const ChannelPlan* __restrict__ plans);
hipError_t set_kernel_attrs_front();

// Agent-scope (write-through / cache-bypassing) accesses for the few words one workgroup hands to another inside a kernel.
template <class T>
__device__ __forceinline__ void agent_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ __forceinline__ T agent_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// k_decide phase 1 for ONE block by the sixteen lanes `sub` = 0..15 of a wave's first lane group (every lane of the group
// calls; lane s reads probe slot s's size): ref lac/encoder.cpp:347-353.
__device__ __forceinline__ void decide_probed_block(uint32_t blk, int sub, BlockPlan* __restrict__ bplans,
                                                    uint32_t* __restrict__ need_full, const ChannelPlan* __restrict__ plans) {
    const uint32_t bytes = sub >= 4 ? agent_load(&plans[(size_t)blk * kSlotsPerBlock + sub].payload_bytes) : 0u;
    const bool is_ms = (sub & 3) >= 2;  // slot = window * 4 + channel, channels L R M S
    uint32_t lr = is_ms ? 0u : bytes, ms = is_ms ? bytes : 0u;  // sums of <= 12 sizes below 2^18: 32 bits
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        lr += (uint32_t)__shfl_xor((int)lr, d, 64);
        ms += (uint32_t)__shfl_xor((int)ms, d, 64);
    }
    if (sub != 0) return;
    const uint8_t choose_ms = ms < lr;
    bplans[blk].choose_ms = choose_ms;
    need_full[blk] = choose_ms ? 0xCu : 0x3u;
}

// k_emit.hip
hipError_t set_kernel_attrs_emit();

// k_analyze.hip: the opt-in to more than 64 KiB of dynamic LDS of every kernel that needs it, once per device
hipError_t ensure_kernel_attrs();

}  // namespace lacx
