// Drop-in mirror of the reference's Block::Encoder (src/codec/block/encoder.hpp:9-30): same class name,
// constructor and methods, implemented over the C ABI of liblacx.so (include/lacx.h).  The analysis runs
// in HIP kernels on an MI355X; errors surface as the exception types the reference's callers expect.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "codec/block/constants.hpp"
#include "lacx.h"

namespace Block {

class Encoder {
public:
    explicit Encoder(int order, bool debug_lpc = false, bool debug_zr = false)
        : order(order), debug_lpc(debug_lpc), debug_zr(debug_zr) {}
    ~Encoder() { reset(); }
    // Copyable like the reference's class: a copy takes the settings and creates its own device handle lazily.
    Encoder(const Encoder& o)
        : order(o.order), debug_lpc(o.debug_lpc), debug_zr(o.debug_zr), zero_run_enabled(o.zero_run_enabled),
          partitioning_enabled(o.partitioning_enabled), debug_partitions(o.debug_partitions),
          block_index(o.block_index), enc(nullptr) {}
    Encoder& operator=(const Encoder& o) {
        if (this != &o) {
            reset();
            order = o.order;
            debug_lpc = o.debug_lpc;
            debug_zr = o.debug_zr;
            zero_run_enabled = o.zero_run_enabled;
            partitioning_enabled = o.partitioning_enabled;
            debug_partitions = o.debug_partitions;
            block_index = o.block_index;
        }
        return *this;
    }

    // input: int32_t PCM block; output: compressed block as bytes (ref block/encoder.cpp:313-838)
    std::vector<uint8_t> encode(const std::vector<int32_t>& pcm) {
        lacx_encoder* h = handle();
        uint8_t* out = nullptr;
        uint64_t size = 0;
        const int rc = lacx_block_encode(h, pcm.data(), static_cast<uint32_t>(pcm.size()), &out, &size);
        if (rc == LACX_E_INVALID) throw std::invalid_argument(lacx_last_error(h));
        if (rc != LACX_OK) throw std::runtime_error(lacx_last_error(h));
        std::vector<uint8_t> bytes(out, out + size);
        lacx_free(out);
        return bytes;
    }

    void set_zero_run_enabled(bool enabled) {
        if (zero_run_enabled != enabled) reset();
        zero_run_enabled = enabled;
    }
    void set_debug_block_index(size_t index) { block_index = index; }
    void set_partitioning_enabled(bool enabled) {
        if (partitioning_enabled != enabled) reset();
        partitioning_enabled = enabled;
    }
    void set_debug_partitions(bool enabled) { debug_partitions = enabled; }

private:
    lacx_encoder* handle() {
        if (!enc) {
            lacx_config cfg{};
            cfg.sample_rate = 48000;
            cfg.bit_depth = 24;
            cfg.stereo_mode = 0;
            cfg.zero_run_enabled = zero_run_enabled;
            cfg.partitioning_enabled = partitioning_enabled;
            cfg.device = -1;
            cfg.emit_threads = 1;
            if (lacx_encoder_create(&cfg, &enc) != LACX_OK) throw std::runtime_error("lacx_encoder_create failed");
        }
        return enc;
    }
    void reset() {
        if (enc) lacx_encoder_destroy(enc);
        enc = nullptr;
    }

    int order;  // stored and ignored, as in the reference (block/encoder.cpp:41)
    bool debug_lpc;
    bool debug_zr;
    bool zero_run_enabled = true;
    bool partitioning_enabled = true;
    bool debug_partitions = false;
    size_t block_index = 0;
    lacx_encoder* enc = nullptr;
};

}  // namespace Block
