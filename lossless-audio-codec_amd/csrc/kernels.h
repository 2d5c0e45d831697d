// kernels.h -- host-visible interface of the kernel translation units (k_front.hip, k_analyze.hip, k_emit.hip,
// decode.hip, wide.hip).
#pragma once
#include <cstdlib>
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
    uint32_t* front_ctr = nullptr; // [num_blocks][2], zero between calls: ingest workgroups / probe slots of the block that are done (kernels_internal.h)
    uint32_t* work_ctr = nullptr;  // a zeroed word: the whole-block analysis runs as persistent workgroups that take their work from it
    // Emit fused into the whole-block analysis kernel (shard-wide arrays, indexed by stream index = block * channels +
    // channel; NOT advanced per pipeline chunk): one fixed-stride staging slot per channel block and a "bitstream is in
    // its slot" flag.
    uint8_t* slots = nullptr;
    unsigned long long slot_stride = 0;
    uint32_t* emitted = nullptr;              // 2: the analysis kernel put the bitstream into the slot
    uint32_t* packed = nullptr;               // 1: the streaming packer moved it to the payload
    unsigned long long* size_rec = nullptr;   // hand-off words of the fused emit (see k_analyze.hip)
    unsigned long long* ready_rec = nullptr;
    unsigned long long* stream_pre = nullptr; // [streams] k_offsets: payload prefix at every stream's first block (sets of several streams)
};

// The whole-block analysis runs as persistent workgroups (one per CU, work taken from counters) where the caller provides
// the zeroed counters.  The streaming packer must then be resident BEFORE that kernel starts: persistent workgroups never
// retire, and a packer that arrives after them finds no CU until the analysis is over (its workgroups are dealt to XCDs
// and shader engines round-robin, whether or not a CU is free there).
inline bool analysis_is_persistent(const DeviceWorkspace& ws) { return ws.work_ctr != nullptr; }

// Launch-shape tuning (read from the environment once, when the encoder is created: Knobs in encoder_impl.h).
struct LaunchTuning {
    uint32_t persistent_grid = 0;  // workgroups of the persistent whole-block analysis (0 = one per CU)
    int pack_nap = 0;              // the streaming packer's polling pause (0 = default)
    int pack_grid = 0;             // the streaming packer's workgroups (0 = default)
    bool fold_front = true;        // the block's stereo estimate by its last ingest workgroup, its LR/MS choice by its last probe slot (LACX_NO_FRONT_FOLD: k_stereo / k_decide as kernels)
    bool no_pairs = false;         // persistent analysis: hand every slot out singly (LACX_NO_PAIRS; A/B of the pair units)
    // Front kernels (ingest, stereo, Levinson, probes, decision) on a stream of their own: `stream` of launch_analysis then
    // carries only the whole-block kernel and what follows, ordered behind the front by an event.  Used by the upload
    // pipeline with a high-priority front stream: the next chunk's front kernels -- short, latency-bound -- then take the
    // CUs that the current chunk's whole-block workgroups free one by one instead of waiting for the end of that kernel.
    hipStream_t front_stream = nullptr;
    // One stream's front kernels in two block halves (one launch set = one stream, >= 1024 blocks): the first half on the
    // launch's stream, the second on `aux_stream` (lower priority, so that it starts when the first half's ingest kernel
    // has been dispatched): the Levinson kernel of a half -- one wave per SIMD, latency-bound, the chip nearly idle -- then
    // runs beside the other half's ingest / probe kernels instead of alone.  aux_ev: two events of the caller's.
    hipStream_t aux_stream = nullptr;
    hipEvent_t aux_ev[2] = {nullptr, nullptr};
};

// Progress reporting of the streaming packer for a device destination that the host drains with a copy engine while
// the analysis runs (one stream).  host_end == nullptr: off.
struct RangeProgress {
    uint32_t* range_cnt = nullptr;            // [ranges] device, zeroed per call: indices of the range in place
    unsigned long long* range_end = nullptr;  // [ranges] device: end offset of the range's last index
    unsigned long long* host_end = nullptr;   // [ranges] pinned host (device address), zeroed per call: end offset + 1 once complete
    uint32_t fuse_total = 0;                  // indices that take part (the ranges cover [0, fuse_total))
    uint32_t fence_mode = 0;                  // diagnostic: 0 system-scope write-back, 1 agent scope, 2 none
};
constexpr uint32_t kPackerRangeItems = 256;

// One launch set on the host: the kernel argument plus what the launchers themselves need to know.
struct LaunchSet {
    BatchRef br;
    const StreamDesc* streams = nullptr;  // host copy of the descriptors (nstreams entries; &br.single for one stream)
    uint32_t nstreams = 0;
    uint32_t total_items = 0;             // channel blocks of the set = its stream indices
    const uint16_t* item_stream = nullptr;  // device: the stream of every stream index (null for one stream)
};
// A set of one stream: the descriptor travels in the kernel arguments.
inline LaunchSet single_set(const StreamDesc& sd) {
    LaunchSet ls;
    ls.br.table = nullptr;
    ls.br.nstreams = 1;
    ls.br.total_blocks = sd.prm.num_blocks;
    ls.br.single = sd;
    ls.nstreams = 1;
    ls.total_items = sd.prm.num_blocks * (sd.prm.channels == 2 ? 2u : 1u);
    return ls;  // (the caller points ls.streams at ls.br.single once the set has its final address)
}

// Where the whole-block analysis kernel writes a channel block's bitstream as soon as its plan is final (slots == null:
// no fused emit; the bitstream then comes from k_offsets + k_emit alone).
// The finished channel block of a slot of nothing but zeros (digital silence): its plan and its bitstream are the same
// for every such slot of the same length under the same settings, so the first workgroup that finishes one leaves them
// here and every later one copies them instead of analysing and emitting again (k_analyze.hip).  Cleared with the other
// per-call words; state: 0 empty, 1 being filled, otherwise the key (length and settings) of the slot it holds.
constexpr uint32_t kSilentBytes = 4096;
struct alignas(16) SilentTemplate {
    uint32_t state;
    uint32_t nbytes;      // bytes of the channel block
    uint32_t plan_words;  // 32-bit words of the plan record in use
    uint32_t pad;
    uint32_t plan[(sizeof(ChannelPlan) + 3) / 4];
    alignas(16) uint8_t bytes[kSilentBytes];
};

struct FuseArgs {
    SilentTemplate* silent = nullptr;      // null: every silent slot is analysed like any other
    uint32_t* silent_copies = nullptr;     // counts the slots that were copies of it
    uint8_t* slots = nullptr;
    unsigned long long slot_stride = 0;
    uint32_t* emitted = nullptr;            // per stream index: 2 = bitstream is in its slot, 0 = not emitted
    uint32_t* err_flag = nullptr;
    unsigned long long* size_rec = nullptr;   // per stream index: 1 << 62 | ms << 61 | bytes, once the plan is final
    unsigned long long* ready_rec = nullptr;  // per stream index: 1 slot published, 2 nothing will come
};

// The small results the host reads after a call (block plans, block table, totals, flags, time stamps) go back in ONE
// kernel that stores them into pinned host memory, instead of one copy engine job each (about 5 us apiece, serialised).
// Sizes in 32-bit words; dst are device-visible addresses of pinned host memory.
struct GatherList {
    static constexpr int kMax = 10;
    const void* src[kMax];
    void* dst[kMax];
    uint32_t words[kMax];
    int n = 0;
    void add(const void* s, void* d, size_t bytes) {
        src[n] = s;
        dst[n] = d;
        words[n] = (uint32_t)(bytes / 4);
        ++n;
    }
};
hipError_t launch_gather(const GatherList& g, hipStream_t stream);

// Enqueues the whole analysis pipeline for one launch set on `stream` (no host synchronisation).
// ev: optional 5 events recorded at: start, after ingest+levinson, after probes+decide, after the
// whole-block analysis kernel, end.  wait_before_full: optional event the whole-block analysis kernel waits for.
hipError_t launch_analysis(const LaunchSet& ls, const DeviceWorkspace& ws, hipStream_t stream, hipEvent_t* ev,
                           const FuseArgs* fuse = nullptr, hipEvent_t wait_before_full = nullptr,
                           const LaunchTuning& tune = LaunchTuning{});

// Device-side emit of the analysed blocks of one set into the result buffer `out` (every stream at its
// StreamDesc::out_base): k_offsets (block byte offsets), k_pack (channel blocks the fused emit left in their staging
// slots: ws.emitted set) and k_emit (all others).
// skip_emitted = false: k_emit emits everything (re-emit into a regrown buffer).
// moved_total / shard_items: the streaming packer's count of channel blocks it has put in place and the number the shard
// has; when they agree k_pack and k_emit have nothing to do and return at once (null: always look).
// wait_before_pack: the packer's completion event; k_offsets does not depend on it and runs in front of the wait.
// repacked (nullable): counts the channel blocks k_pack had to move.
// lazy_repair: only k_offsets and the wait for the packer are enqueued; whether k_pack / k_emit have anything to do the
// host sees from the packer's count of moved channel blocks (gathered with the rest) and enqueues them in the rare case
// that they do -- on the common path two kernels and an event less between the end of the analysis and the host's wake-up.
hipError_t launch_emit(const LaunchSet& ls, const DeviceWorkspace& ws, uint8_t* out,
                       const unsigned long long* base_ptr, hipEvent_t wait_before_offsets,
                       hipEvent_t offsets_done, hipStream_t stream, bool skip_emitted = true,
                       const uint32_t* moved_total = nullptr, uint32_t shard_items = 0,
                       hipEvent_t wait_before_pack = nullptr, uint32_t* repacked = nullptr, bool lazy_repair = false);

// The streaming packer of the fused emit: runs beside the analysis kernels on its own stream and moves the staging
// slots of the set's stream indices to their place in `out` as they are published.  counters: [0] error flags, [1] the
// number of channel blocks it has put in place, [2] packer waves that gave up waiting for a record.
hipError_t launch_stream_out(const LaunchSet& ls, const DeviceWorkspace& ws, uint8_t* out, uint32_t* counters,
                             hipStream_t stream, const RangeProgress& rp = RangeProgress{},
                             const LaunchTuning& tune = LaunchTuning{});

// The decoder (decode.hip): one lane per block; payload must be followed by kDecodeTailPad readable zero bytes (the bit
// reader's bounded look-ahead past the last block, see BitIn).
constexpr size_t kDecodeTailPad = 128;
// The decoder's entry points:  byte_off / frame_off:
// [num_blocks + 1] prefix sums of the block table; status[blk] = 0 or an error code, ms_flag[blk] = the block's LR/MS flag.
hipError_t launch_decode(uint32_t num_blocks, int channels, int stereo_mode, int bit_depth, const uint8_t* payload,
                         const unsigned long long* byte_off, const unsigned long long* frame_off, int32_t* left,
                         int32_t* right, uint32_t* status, uint8_t* ms_flag, hipStream_t stream);

// The legacy version-2 container (no compressed block sizes): one lane walks the whole payload.
hipError_t launch_decode_serial(uint32_t num_blocks, int channels, int stereo_mode, int bit_depth, const uint8_t* payload,
                                uint32_t payload_bits, const unsigned long long* frame_off, int32_t* left, int32_t* right,
                                uint32_t* status, uint8_t* ms_flag, hipStream_t stream);

// Block::Encoder::encode's analysis for one block of arbitrary int32 samples (wide.hip): d_res = scratch for the eleven
// candidate residuals ([11][kMaxBlock] int32), d_plan receives the plan.
hipError_t launch_wide_block(const int32_t* d_x, uint32_t n, int zero_run, int partitioning, int32_t* d_res,
                             ChannelPlan* d_plan, hipStream_t stream);

size_t analyze_smem_bytes_full();
// Diagnostic builds (-DLACX_STAMPS) only: per-phase shader-cycle sums over all waves; returns 0 otherwise.
int debug_read_stamps(unsigned long long* out32);
int debug_read_offset_stamps(unsigned long long* out8);  // k_offsets, diagnostic builds (k_emit.hip)
size_t analyze_smem_bytes_probe();

}  // namespace lacx
