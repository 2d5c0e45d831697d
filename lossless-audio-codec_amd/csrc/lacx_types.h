// lacx_types.h -- records exchanged between the HIP kernels and the host side of the encoder.
#pragma once
#include <stdint.h>

namespace lacx {

constexpr int kMaxBlock = 16384;  // Block::MAX_BLOCK_SIZE        (ref src/codec/block/constants.hpp:6)
constexpr int kMaxParts = 256;    // 1 << MAX_PARTITION_ORDER     (ref constants.hpp:11)
constexpr int kMinPartition = 32; // MIN_PARTITION_SIZE           (ref constants.hpp:10)
constexpr int kMaxPartitionOrder = 8;
constexpr int kProbe = 256;             // kStereoProbeSize             (ref src/codec/lac/encoder.cpp:19)
constexpr int kFullCompareLimit = 4096; // kStereoFullComparisonLimit   (ref lac/encoder.cpp:20)

// Channel kinds inside a stereo block.
enum : int { CH_L = 0, CH_R = 1, CH_M = 2, CH_S = 3 };

// A "slot" is one channel-segment that gets the full Block::Encoder analysis:
//   slot = window * 4 + channel,  window 0 = whole block, windows 1..3 = the three 256-frame probes
//   (ref lac/encoder.cpp:343-346).  Mono streams use slot 0 only.
constexpr int kSlotsPerBlock = 16;

// Quantised LPC candidates for one slot: orders {4,6,8,10,12} (ref block/encoder.cpp:41).
struct LpcSet {
    int16_t coef[5][13];  // coef[ci][1..used] valid, rest 0
    uint8_t used[5];      // 0 = candidate skipped
    uint8_t pad;
};

// Everything Block::Encoder::encode decides before it starts emitting bits
// (ref block/encoder.cpp:313-552), for one slot.
struct ChannelPlan {
    uint8_t predictor_type;   // 0 fixed, 1 FIR, 2 LPC
    uint8_t order;            // chosen_order
    uint8_t partition_order;  // 0..8
    uint8_t valid;            // 1 when the slot was analysed
    int16_t coef[12];         // coef[i-1] = Q15 coefficient i (LPC only)
    uint32_t payload_bytes;   // exact size of the emitted channel block
    uint64_t total_bits;      // best_total_bits (metadata + residual bits, padded to a byte)
    uint8_t part_mode_k[kMaxParts];  // (mode << 5) | k per partition
};

// Per-block stereo decision (ref lac/encoder.cpp:126-197, 321-373).
struct BlockPlan {
    uint8_t choose_ms;   // final LR(0)/MS(1) choice
    uint8_t uncertain;   // estimate_stereo_mode's flag
    uint8_t est_ms;      // estimate_stereo_mode's choose_ms
    uint8_t invalid;     // 1 if a sample was outside the bit depth (ref lac/encoder.cpp:82-102)
    uint32_t frames;     // frames in this block
    uint32_t first_bad;  // index (within block) of the first invalid sample, channel in bit 31
    uint32_t pad;
};

struct AnalyzeParams {
    uint64_t frames;       // total frames in the stream segment handed to the kernels
    uint32_t num_blocks;
    uint32_t first_block;  // blocks [first_block, first_block+num_blocks) are processed
    int32_t channels;      // 1 or 2
    int32_t stereo_mode;   // 0 LR, 1 MS, 2 per-block auto
    int32_t bit_depth;     // 16 / 24 (range validation); 0 = no validation (Block::Encoder path)
    int32_t zero_run;
    int32_t partitioning;
    uint32_t debug_skip;   // diagnostic ablation mask (timing experiments only; 0 in production)
    int32_t layout;        // PCM_PLANAR_I32 / PCM_INTERLEAVED_I16 / PCM_INTERLEAVED_I24 (analyze_core.h)
    uint32_t stream_base;  // fused emit: stream index (block * channels + channel) of the chunk's first channel block
};

// One input stream of a launch set.  A launch set covers the blocks of one or many streams (lacx_encode_batch: the files
// of a corpus as ONE job, ref src/codec/lac/encoder.cpp:404-435 keeps one pool over all blocks): global block g of the
// set belongs to the stream with first_block <= g < first_block + prm.num_blocks and is block g - first_block of it.
// Workspace arrays (plans, need masks, autocorrelations ...) are indexed by the global block, sample addresses and block
// geometry by the stream's own block number.
struct StreamDesc {
    AnalyzeParams prm;              // the stream's parameters; prm.stream_base = its first stream index (block * channels + channel, over the set)
    const int32_t* left;            // planar: left channel; interleaved layouts: the WAV data chunk
    const int32_t* right;           // planar: right channel (null for mono)
    uint32_t first_block;           // first global block
    uint32_t first_wg;              // first workgroup in the grid of the whole-block analysis kernel (channels per block)
    uint32_t fuse_items;            // stream indices [prm.stream_base, prm.stream_base + fuse_items) take part in the fused emit
    uint32_t pad;                   // the stream's number in a table of several
    unsigned long long out_base;    // byte offset of the stream's payload region in the result buffer
    unsigned long long out_cap;     // bytes reserved for it
};

// What every kernel gets: a table of streams in device memory, or -- one stream, the common case -- the descriptor
// itself in the kernel arguments (table == nullptr; no upload, no look-up).
struct BatchRef {
    const StreamDesc* table;
    uint32_t nstreams;
    uint32_t total_blocks;
    StreamDesc single;
};

}  // namespace lacx
