// kernels.h -- host-visible interface of kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>

#include "lacx_types.h"

namespace lacx {

// Device buffers sized for `num_blocks` blocks (allocated by the API layer).
struct DeviceWorkspace {
    ChannelPlan* plans = nullptr;   // [num_blocks][kSlotsPerBlock]
    BlockPlan* bplans = nullptr;    // [num_blocks]
    uint32_t* need_probe = nullptr; // [num_blocks] slot masks
    uint32_t* need_full = nullptr;  // [num_blocks]
    int64_t* acorr = nullptr;       // [num_blocks][kSlotsPerBlock][13]
    LpcSet* lpcs = nullptr;         // [num_blocks][kSlotsPerBlock]
    unsigned long long* sums = nullptr;  // [num_blocks][12] stereo proxy sums
    uint32_t* badidx = nullptr;     // [num_blocks][2] first out-of-range sample per channel
    unsigned long long* block_off = nullptr;  // [num_blocks + 1] payload byte offsets (device emit)
    uint32_t* table = nullptr;      // [num_blocks][2] (frames, bytes) block table entries
    uint32_t* err_flag = nullptr;   // device emit consistency flag
    // device clock (100 MHz) at which the first / last workgroup of the whole-block analysis kernel started /
    // ended: its execution span without the time it queued behind other streams (null = not recorded)
    unsigned long long* t_first = nullptr;
    unsigned long long* t_last = nullptr;
};

// Enqueues the whole analysis pipeline for one shard on `stream` (no host synchronisation).
// ev: optional 5 events recorded at: start, after ingest+levinson, after probes+decide, after the
// whole-block analysis kernel, end.
hipError_t launch_analysis(const int32_t* d_left, const int32_t* d_right, const AnalyzeParams& prm,
                           const DeviceWorkspace& ws, hipStream_t stream, hipEvent_t* ev);

// Device-side emit of the analysed blocks of one chunk into d_payload (k_offsets + k_emit).
hipError_t launch_emit(const int32_t* d_left, const int32_t* d_right, const AnalyzeParams& prm,
                       const DeviceWorkspace& ws, uint8_t* out, unsigned long long out_cap,
                       const unsigned long long* base_ptr, hipEvent_t wait_before_offsets,
                       hipEvent_t offsets_done, hipStream_t stream);

size_t analyze_smem_bytes_full();
// Diagnostic builds (-DLACX_STAMPS) only: per-phase shader-cycle sums over all waves; returns 0 otherwise.
int debug_read_stamps(unsigned long long* out32);
size_t analyze_smem_bytes_probe();

}  // namespace lacx
