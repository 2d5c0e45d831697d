// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A thin extern "C" wrapper around the *unmodified* reference sources as they lie under
// /root/reference (compiled in place by oracle/Makefile into oracle/_ref/liblac_ref.so; nothing from
// the reference is copied into this repository).  It exists so that tests/ and the fixture-minting
// script can (a) validate the C restatement in oracle/lac_oracle.c and (b) mint golden vectors.
// It is never linked into, loaded by, or shipped with the product library.
//
// Wrapped reference entry points:
//   LAC::Encoder::encode         src/codec/lac/encoder.hpp:12-43, encoder.cpp:215
//   LAC::Decoder::decode         src/codec/lac/decoder.hpp:10-24
//   Block::Encoder::encode       src/codec/block/encoder.hpp:9-30, encoder.cpp:313
//   LPC::analyze_block_q15       src/codec/lpc/lpc.hpp:11-14
//   Rice::adapt_k                src/codec/rice/rice.hpp:38
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>
#include <vector>

#include "codec/block/encoder.hpp"
#include "codec/lac/decoder.hpp"
#include "codec/lac/encoder.hpp"
#include "codec/lpc/lpc.hpp"
#include "codec/rice/rice.hpp"
#include "io/wav_io.hpp"

namespace {
thread_local std::string g_last_error;

uint8_t* dup_bytes(const std::vector<uint8_t>& v, uint64_t* out_size) {
    *out_size = v.size();
    uint8_t* p = static_cast<uint8_t*>(std::malloc(v.size() ? v.size() : 1));
    if (p && !v.empty()) std::memcpy(p, v.data(), v.size());
    return p;
}
}  // namespace

extern "C" {

const char* lacref_last_error() { return g_last_error.c_str(); }

void lacref_free(void* p) { std::free(p); }

// Returns 0 ok, 1 std::invalid_argument, 2 std::runtime_error, 3 other.
int lacref_encode(const int32_t* left, const int32_t* right, uint64_t frames, int channels,
                  uint32_t sample_rate, int bit_depth, int stereo_mode, int zero_run, int partitioning,
                  int threads, uint8_t** out, uint64_t* out_size) {
    try {
        std::vector<int32_t> l(left, left + frames);
        std::vector<int32_t> r;
        if (channels == 2) r.assign(right, right + frames);
        LAC::Encoder enc(12, static_cast<uint8_t>(stereo_mode), sample_rate,
                         static_cast<uint8_t>(bit_depth));
        enc.set_zero_run_enabled(zero_run != 0);
        enc.set_partitioning_enabled(partitioning != 0);
        enc.set_thread_count(static_cast<size_t>(threads));
        std::vector<uint8_t> bytes = enc.encode(l, r);
        *out = dup_bytes(bytes, out_size);
        return 0;
    } catch (const std::invalid_argument& e) {
        g_last_error = e.what();
        return 1;
    } catch (const std::runtime_error& e) {
        g_last_error = e.what();
        return 2;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return 3;
    }
}

// Decodes a .lac image; returns 0 on success. left/right are malloc'd int32 arrays (right NULL if mono).
int lacref_decode(const uint8_t* data, uint64_t size, int32_t** left, int32_t** right, uint64_t* frames,
                  int* channels, uint32_t* sample_rate, int* bit_depth, int* stereo_mode) {
    try {
        LAC::Decoder dec;
        std::vector<int32_t> l, r;
        FrameHeader hdr;
        dec.decode(data, size, l, r, &hdr);
        *frames = l.size();
        *channels = hdr.channels;
        *sample_rate = hdr.sample_rate;
        *bit_depth = hdr.bit_depth;
        *stereo_mode = hdr.stereo_mode;
        *left = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (l.size() ? l.size() : 1)));
        std::memcpy(*left, l.data(), sizeof(int32_t) * l.size());
        if (hdr.channels == 2) {
            *right = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (r.size() ? r.size() : 1)));
            std::memcpy(*right, r.data(), sizeof(int32_t) * r.size());
        } else {
            *right = nullptr;
        }
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return 1;
    }
}

int lacref_block_encode(const int32_t* pcm, uint32_t n, int zero_run, int partitioning, uint8_t** out,
                        uint64_t* out_size) {
    try {
        std::vector<int32_t> v(pcm, pcm + n);
        Block::Encoder enc(12);
        enc.set_zero_run_enabled(zero_run != 0);
        enc.set_partitioning_enabled(partitioning != 0);
        std::vector<uint8_t> bytes = enc.encode(v);
        *out = dup_bytes(bytes, out_size);
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return 1;
    }
}

// coeffs_q15 must hold order+1 entries.  Returns used_order.
int lacref_lpc_analyze(const int32_t* pcm, uint32_t n, int order, int16_t* coeffs_q15) {
    std::vector<int32_t> v(pcm, pcm + n);
    LPC lpc(order);
    std::vector<int16_t> c;
    int used = 0;
    lpc.analyze_block_q15(v, c, used, nullptr);
    for (int i = 0; i <= order; ++i) coeffs_q15[i] = c[static_cast<size_t>(i)];
    return used;
}

// Runs Rice::adapt_k over a sequence of unsigned values; k_out[i] is the k returned after value i.
void lacref_adapt_k_sequence(const uint32_t* u, uint32_t n, uint32_t* k_out) {
    Rice::AdaptState st;
    uint64_t sum = 0;
    for (uint32_t i = 0; i < n; ++i) {
        sum += u[i];
        k_out[i] = Rice::adapt_k(sum, i + 1, st);
    }
}

// read_wav (ref src/io/wav_io.cpp:167-277) on a file path.  Returns 1 when the reference accepts the file and
// fills the format fields; left/right (caller buffers of `cap` samples each, may be NULL) receive the samples.
int lacref_read_wav(const char* path, uint16_t* channels, uint32_t* sample_rate, uint8_t* bit_depth, uint64_t* frames,
                    int32_t* left, int32_t* right, uint64_t cap) {
    std::vector<int32_t> l, r;
    uint16_t ch = 0;
    uint32_t sr = 0;
    uint8_t bd = 0;
    if (!read_wav(path, l, r, ch, sr, bd)) return 0;
    *channels = ch;
    *sample_rate = sr;
    *bit_depth = bd;
    *frames = l.size();
    if (left && l.size() <= cap) std::memcpy(left, l.data(), l.size() * sizeof(int32_t));
    if (right && r.size() <= cap && !r.empty()) std::memcpy(right, r.data(), r.size() * sizeof(int32_t));
    return 1;
}

}  // extern "C"
