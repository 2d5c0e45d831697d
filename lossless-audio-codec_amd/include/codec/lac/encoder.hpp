// Drop-in mirror of the reference's LAC::Encoder (src/codec/lac/encoder.hpp:12-43): same constructor,
// encode() and setters, implemented over the C ABI of liblacx.so (include/lacx.h).  Exceptions follow the
// reference: std::invalid_argument for bad arguments / out-of-range samples
// (src/codec/lac/encoder.cpp:220-241), std::runtime_error otherwise (:447-449; also device failures).
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "codec/block/encoder.hpp"
#include "codec/lac/thread_collector.hpp"
#include "lacx.h"

namespace LAC {

class Encoder {
public:
    Encoder(uint8_t order, uint8_t stereo_mode = 0, uint32_t sample_rate = 44100, uint8_t bit_depth = 16,
            bool debug_lpc = false, bool debug_stereo_est = false, bool debug_zr = false)
        : order(order), stereo_mode(stereo_mode), sample_rate(sample_rate), bit_depth(bit_depth), debug_lpc(debug_lpc),
          debug_stereo_est(debug_stereo_est), debug_zr(debug_zr) {}
    ~Encoder() { reset(); }
    // Copyable like the reference's class (a plain aggregate of settings there): a copy takes the settings and
    // creates its own device handle lazily, on its first encode.
    Encoder(const Encoder& o)
        : order(o.order), stereo_mode(o.stereo_mode), sample_rate(o.sample_rate), bit_depth(o.bit_depth),
          debug_lpc(o.debug_lpc), debug_stereo_est(o.debug_stereo_est), debug_zr(o.debug_zr),
          zero_run_enabled(o.zero_run_enabled), partitioning_enabled(o.partitioning_enabled),
          debug_partitions(o.debug_partitions), thread_count(o.thread_count), devices(o.devices),
          min_blocks_per_device(o.min_blocks_per_device), enc(nullptr) {}
    Encoder& operator=(const Encoder& o) {
        if (this != &o) {
            reset();
            order = o.order;
            stereo_mode = o.stereo_mode;
            sample_rate = o.sample_rate;
            bit_depth = o.bit_depth;
            debug_lpc = o.debug_lpc;
            debug_stereo_est = o.debug_stereo_est;
            debug_zr = o.debug_zr;
            zero_run_enabled = o.zero_run_enabled;
            partitioning_enabled = o.partitioning_enabled;
            debug_partitions = o.debug_partitions;
            thread_count = o.thread_count;
            devices = o.devices;
            min_blocks_per_device = o.min_blocks_per_device;
        }
        return *this;
    }

    std::vector<uint8_t> encode(const std::vector<int32_t>& left, const std::vector<int32_t>& right,
                                ThreadCollector* collector = nullptr) {
        if (left.empty()) throw std::invalid_argument("left channel must not be empty");
        if (!right.empty() && right.size() != left.size()) {
            throw std::invalid_argument("right channel size (" + std::to_string(right.size()) +
                                        ") must match left channel size (" + std::to_string(left.size()) + ")");
        }
        if (collector) collector->record(std::this_thread::get_id());
        lacx_encoder* h = handle();
        uint8_t* out = nullptr;
        uint64_t size = 0;
        const int rc = lacx_encode(h, left.data(), right.empty() ? nullptr : right.data(), left.size(), &out, &size);
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
    void set_partitioning_enabled(bool enabled) {
        if (partitioning_enabled != enabled) reset();
        partitioning_enabled = enabled;
    }
    void set_debug_partitions(bool enabled) { debug_partitions = enabled; }
    void set_thread_count(size_t max_threads) {  // host emit workers here (0 = auto)
        if (thread_count != max_threads) reset();
        thread_count = max_threads;
    }
    // Not in the reference: the devices encode() spreads the stream's blocks over, the way the reference spreads them over
    // its worker threads (src/codec/lac/encoder.cpp:385-443).  Default (empty list): every visible device; a stream is
    // spread over fewer when a device would get fewer than min_blocks blocks (0 = the library's default, 64).
    void set_devices(const std::vector<int>& list, uint32_t min_blocks = 0) {
        reset();
        devices.assign(list.begin(), list.end());
        min_blocks_per_device = min_blocks;
    }

private:
    lacx_encoder* handle() {
        if (!enc) {
            lacx_config cfg{};
            cfg.sample_rate = sample_rate;
            cfg.bit_depth = bit_depth;
            cfg.stereo_mode = stereo_mode;
            cfg.zero_run_enabled = zero_run_enabled;
            cfg.partitioning_enabled = partitioning_enabled;
            cfg.device = LACX_DEVICE_ALL;
            cfg.emit_threads = static_cast<uint32_t>(thread_count);
            const int rc = devices.empty() ? lacx_encoder_create(&cfg, &enc)
                                           : lacx_encoder_create_multi(&cfg, devices.data(), static_cast<uint32_t>(devices.size()),
                                                                       min_blocks_per_device, &enc);
            if (rc != LACX_OK) throw std::runtime_error("lacx_encoder_create failed");
        }
        return enc;
    }
    void reset() {
        if (enc) lacx_encoder_destroy(enc);
        enc = nullptr;
    }

    uint8_t order;  // stored and ignored, as in the reference
    uint8_t stereo_mode;
    uint32_t sample_rate;
    uint8_t bit_depth;
    bool debug_lpc;
    bool debug_stereo_est;
    bool debug_zr;
    bool zero_run_enabled = true;
    bool partitioning_enabled = true;
    bool debug_partitions = false;
    size_t thread_count = 0;
    std::vector<int32_t> devices;
    uint32_t min_blocks_per_device = 0;
    lacx_encoder* enc = nullptr;
};

}  // namespace LAC
