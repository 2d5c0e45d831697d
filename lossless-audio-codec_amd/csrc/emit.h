// emit.h -- host side of the encoder: the bit-serial Rice / zero-run / bin emit that BASELINE.json's
// north_star leaves on the host, driven by the ChannelPlan records the kernels produce, and the v3
// container writer.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "lacx_types.h"

namespace lacx {

// Emits one channel block (ref src/codec/block/encoder.cpp:554-838) into out[0..plan.payload_bytes).
// `a`/`b`/`kind` describe the samples as in SlotSrc (kind CH_M / CH_S derive mid/side from a=left,
// b=right).  Returns the number of bytes written, which always equals plan.payload_bytes when the
// plan is consistent with the samples; (size_t)-1 if the buffer would overflow (inconsistent plan).
size_t emit_channel(const ChannelPlan& plan, const int32_t* a, const int32_t* b, int kind, uint32_t n,
                    uint8_t* out, size_t cap, int32_t* scratch /* n int32 */);

struct StreamParams {
    uint32_t sample_rate;
    uint8_t bit_depth;
    uint8_t channels;
    uint8_t stereo_mode;  // header value: 0 for mono
};

// Bytes of one block's payload: [flag byte if per-block stereo] + the chosen channel payloads.
uint32_t block_payload_bytes(const StreamParams& sp, const BlockPlan& bp, const ChannelPlan* slots);

// Emits the payloads of blocks [0, nblocks) of a shard into `payload` (sized by the caller from
// block_payload_bytes) using `threads` worker threads. offsets[b] = byte offset of block b.
// Returns an empty string on success, else an error description.
std::string emit_blocks(const StreamParams& sp, const int32_t* left, const int32_t* right, uint64_t frames,
                        const BlockPlan* bplans, const ChannelPlan* plans, uint32_t nblocks,
                        const uint64_t* offsets, uint8_t* payload, uint64_t payload_size, unsigned threads);

// Emits block `b` ([flag] + channel payloads) into out[0..cap); cap must equal block_payload_bytes.
// Returns false when the emitted size disagrees with the plan.
bool emit_one_block(const StreamParams& sp, const int32_t* left, const int32_t* right, uint64_t frames,
                    const BlockPlan& bp, const ChannelPlan* slots, uint32_t b, uint8_t* out, size_t cap,
                    int32_t* scratch /* kMaxBlock int32 */);

// Persistent emit workers.  A job covers blocks [0, nblocks); a block becomes eligible once
// publish(upto) has been called with upto > block (its plan records and byte offset are then final),
// which lets the emit of early blocks overlap the device analysis of later ones.
class EmitPool {
public:
    explicit EmitPool(unsigned threads);
    ~EmitPool();
    EmitPool(const EmitPool&) = delete;
    EmitPool& operator=(const EmitPool&) = delete;

    unsigned threads() const;
    // offsets has nblocks + 1 entries, filled in by the caller before the matching publish().
    void begin(const StreamParams& sp, const int32_t* left, const int32_t* right, uint64_t frames,
               const BlockPlan* bplans, const ChannelPlan* plans, uint32_t nblocks, const uint64_t* offsets,
               uint8_t* payload);
    void publish(uint32_t ready_upto);
    void abort();
    bool finish();  // waits for every block; false if any block failed or the job was aborted

private:
    struct Impl;
    Impl* impl_;
};

// 10-byte frame header (ref src/codec/frame/frame_header.hpp:25-36).
void write_frame_header(const StreamParams& sp, uint8_t* out10);

}  // namespace lacx
