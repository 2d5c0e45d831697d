// RIFF/WAVE container walk over a memory buffer: the host half of the interleaved ingest.
// Accepts and rejects exactly the files the reference's read_wav does (ref src/io/wav_io.cpp:167-277):
//   "RIFF" <size == file size - 8> "WAVE"; chunks with even padding that must fit the file;
//   one 16-byte "fmt " before "data": PCM (format 1), 16/24 bits, 44100/48000/96000/192000 Hz, 1-2 channels,
//   block_align and byte_rate consistent; one non-empty "data" chunk, a whole number of frames, at most
//   1 GiB once widened to int32; every other chunk is skipped; both chunks must be present.
// Instead of widening the samples on the host (the reference's per-sample loop at :248-261) the caller gets
// the position of the raw data chunk, which the kernels read as it is.
#pragma once
#include <cstdint>
#include <cstring>

namespace lacx {

struct WavInfo {
    uint16_t channels = 0;
    uint16_t bit_depth = 0;
    uint32_t sample_rate = 0;
    uint64_t frames = 0;
    uint64_t data_offset = 0;
    uint64_t data_bytes = 0;
};

inline bool wav_parse(const uint8_t* p, uint64_t size, WavInfo* out) {
    auto u16 = [&](uint64_t o) { return (uint16_t)(p[o] | (p[o + 1] << 8)); };
    auto u32 = [&](uint64_t o) {
        return (uint32_t)p[o] | ((uint32_t)p[o + 1] << 8) | ((uint32_t)p[o + 2] << 16) | ((uint32_t)p[o + 3] << 24);
    };
    if (!p || size < 12) return false;
    if (std::memcmp(p, "RIFF", 4) != 0) return false;
    if ((uint64_t)u32(4) + 8u != size) return false;
    if (std::memcmp(p + 8, "WAVE", 4) != 0) return false;
    bool got_fmt = false, got_data = false;
    WavInfo w;
    uint16_t block_align = 0;
    uint64_t pos = 12, remaining = size - 12;
    while (remaining > 0) {
        if (remaining < 8) return false;
        const uint8_t* id = p + pos;
        const uint32_t chunk = u32(pos + 4);
        pos += 8;
        remaining -= 8;
        const uint64_t padded = (uint64_t)chunk + (chunk & 1u);
        if (padded > remaining) return false;
        if (std::memcmp(id, "fmt ", 4) == 0) {
            if (got_fmt || got_data || chunk != 16u) return false;
            const uint16_t format = u16(pos), channels = u16(pos + 2);
            const uint32_t rate = u32(pos + 4), byte_rate = u32(pos + 8);
            const uint16_t align = u16(pos + 12), bits = u16(pos + 14);
            if (format != 1) return false;
            if (bits != 16 && bits != 24) return false;
            if (rate != 44100 && rate != 48000 && rate != 96000 && rate != 192000) return false;
            if (channels != 1 && channels != 2) return false;
            const uint16_t expect = (uint16_t)(channels * (bits / 8));
            if (align != expect) return false;
            if (byte_rate != rate * expect) return false;
            w.channels = channels;
            w.bit_depth = bits;
            w.sample_rate = rate;
            block_align = align;
            got_fmt = true;
        } else if (std::memcmp(id, "data", 4) == 0) {
            if (!got_fmt || got_data || chunk == 0u) return false;
            if (chunk % block_align != 0) return false;
            const uint64_t frames = chunk / block_align;
            if (frames * w.channels * 4ull > (1ull << 30)) return false;
            w.frames = frames;
            w.data_offset = pos;
            w.data_bytes = chunk;
            got_data = true;
        }
        pos += padded;
        remaining -= padded;
    }
    if (!got_fmt || !got_data) return false;
    *out = w;
    return true;
}

}  // namespace lacx
