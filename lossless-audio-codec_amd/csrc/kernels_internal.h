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
__global__ void k_ingest(BatchRef br, unsigned long long* __restrict__ sums, uint32_t* __restrict__ badidx,
                         int64_t* __restrict__ acorr);
__global__ void k_stereo(BatchRef br, const unsigned long long* __restrict__ sums, const uint32_t* __restrict__ badidx,
                         BlockPlan* __restrict__ bplans, uint32_t* __restrict__ need_probe, uint32_t* __restrict__ need_full);
__global__ void k_levinson(BatchRef br, const int64_t* __restrict__ acorr, const uint32_t* __restrict__ need_probe,
                           LpcSet* __restrict__ lpcs);
__global__ void k_decide(BatchRef br, int phase, BlockPlan* __restrict__ bplans, const uint32_t* __restrict__ need_probe,
                         uint32_t* __restrict__ need_full, const ChannelPlan* __restrict__ plans);
hipError_t set_kernel_attrs_front();

// k_emit.hip
hipError_t set_kernel_attrs_emit();

// k_analyze.hip: the opt-in to more than 64 KiB of dynamic LDS of every kernel that needs it, once per device
hipError_t ensure_kernel_attrs();

}  // namespace lacx
