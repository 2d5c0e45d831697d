// api_decode.cpp -- the decode entry points of the C ABI: container parsing and the device decoder's host side.
#include "encoder_impl.h"

extern "C" {

// ---- decode -------------------------------------------------------------------------------------
namespace {
thread_local std::string g_decode_err;
int decode_fail(int code, const std::string& msg) {
    g_decode_err = msg;
    return code;
}
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
}  // namespace

const char* lacx_decode_last_error(void) { return g_decode_err.c_str(); }

// Container header + block table: the structural rules of the reference's reader (src/codec/frame/frame_header.hpp:48-74,
// lac/decoder.cpp:84-145) -- sync, version 3, channels, stereo mode (0 for mono), one of the four sample rates, depth,
// reserved byte; at least one block; every block 1..16384 frames, non-final ones at least 256; non-zero compressed
// sizes that add up to the file; at most 6 912 000 000 samples and a WAV that RIFF can hold.  NOT taken over: its cap on
// the decoded PCM (1 GiB) and the block count that follows from it, which would refuse the 2 h stream of BASELINE
// configs[3].  The legacy version-2 container (no compressed sizes, hence no parallelism) is read too: one lane walks it.
int lacx_stream_parse(const uint8_t* lac, uint64_t size, lacx_stream_info* out) {
    if (!lac || !out) return decode_fail(LACX_E_INVALID, "null argument");
    if (size == 0) return decode_fail(LACX_E_INVALID, "[decode-error] empty input");
    if (size < 10 || lac[0] != 0x4C || lac[1] != 0x41 || (lac[2] != 3 && lac[2] != 2))
        return decode_fail(LACX_E_INVALID, "[decode-error] invalid frame header");
    const int version = lac[2], ch = lac[3], sm = lac[4], bd = lac[8];
    const uint32_t sr = ((uint32_t)lac[5] << 8) | lac[6] | ((uint32_t)lac[7] << 16);
    const bool rate_ok = sr == 44100 || sr == 48000 || sr == 96000 || sr == 192000;
    if ((ch != 1 && ch != 2) || sm > 2 || (ch == 1 && sm != 0) || !rate_ok || (bd != 16 && bd != 24) || lac[9] != 0)
        return decode_fail(LACX_E_INVALID, "[decode-error] invalid frame header");
    if (size < 14) return decode_fail(LACX_E_INVALID, "[decode-error] invalid block count");
    const uint32_t nb = be32(lac + 10);
    if (nb == 0) return decode_fail(LACX_E_INVALID, "[decode-error] invalid block count");
    const uint64_t entry = version >= 3 ? 8u : 4u;  // version 2 has no compressed sizes (ref lac/decoder.cpp:100-104)
    if (size < 14 + entry * nb) return decode_fail(LACX_E_INVALID, "[decode-error] truncated block size table");
    uint64_t frames = 0, pay = 0;
    for (uint32_t b = 0; b < nb; ++b) {
        const uint32_t n = be32(lac + 14 + entry * b);
        if (n == 0 || n > (uint32_t)kMaxBlock || (b + 1 < nb && n < 256u)) return decode_fail(LACX_E_INVALID, "[decode-error] invalid block size");
        frames += n;
        if (frames > 6912000000ull) return decode_fail(LACX_E_INVALID, "[decode-error] total samples exceed maximum");
        if (version >= 3) {
            const uint32_t by = be32(lac + 18 + 8ull * b);
            // The device reader's bit positions are 32-bit and relative to the block: a block must stay below 2^29 bytes.
            // (The reference takes any non-zero size that fits the file; a block this long -- a Rice token at k = 0 may
            // carry a unary part of up to 2^30 bits -- is a documented deviation, see lacx.h.)
            if (by == 0 || by >= (1u << 29)) return decode_fail(LACX_E_INVALID, "[decode-error] invalid compressed block size");
            pay += by;
            if (pay > size) return decode_fail(LACX_E_INVALID, "[decode-error] compressed block sizes exceed frame payload");
        }
    }
    const uint64_t wav_bytes = frames * (uint64_t)ch * (uint64_t)(bd / 8);
    if (36u + wav_bytes + (wav_bytes & 1u) > 0xFFFFFFFFull) return decode_fail(LACX_E_INVALID, "[decode-error] decoded WAV data exceeds RIFF limit");
    if (version >= 3 && 14 + 8ull * nb + pay != size) return decode_fail(LACX_E_INVALID, "[decode-error] block payloads do not fill the file");
    if (version == 2 && size - (14 + 4ull * nb) >= (1ull << 29)) return decode_fail(LACX_E_INVALID, "[decode-error] version-2 payload too large for the serial reader");
    out->sample_rate = sr;
    out->blocks = nb;
    out->frames = frames;
    out->channels = (uint8_t)ch;
    out->bit_depth = (uint8_t)bd;
    out->stereo_mode = (uint8_t)sm;
    out->version = (uint8_t)version;
    return LACX_OK;
}

// The decoder object: device buffers, a stream and two events that live from call to call (grow-only), so that a decode
// costs its copies and its kernel, not six allocations (ref LAC::Decoder is an object too, src/codec/lac/decoder.hpp:10-24).
struct lacx_decoder {
    int device = -1;  // -1: whatever device is current at the first call
    bool ready = false;
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    uint8_t* d_pay = nullptr;
    uint64_t pay_cap = 0;
    unsigned long long* d_offs = nullptr;
    uint64_t offs_cap = 0;
    int32_t *d_left = nullptr, *d_right = nullptr;
    uint64_t pcm_cap = 0;
    uint32_t* d_status = nullptr;
    uint8_t* d_ms = nullptr;
    uint32_t* h_status = nullptr;          // pinned
    unsigned long long* h_offs = nullptr;  // pinned
    uint32_t blocks_cap = 0;
    std::string err;
};

namespace {
void decoder_release(lacx_decoder* d) {
    if (d->ready) (void)hipSetDevice(d->device);
    if (d->e0) (void)hipEventDestroy(d->e0);
    if (d->e1) (void)hipEventDestroy(d->e1);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    if (d->d_pay) (void)hipFree(d->d_pay);
    if (d->d_offs) (void)hipFree(d->d_offs);
    if (d->d_left) (void)hipFree(d->d_left);
    if (d->d_right) (void)hipFree(d->d_right);
    if (d->d_status) (void)hipFree(d->d_status);
    if (d->d_ms) (void)hipFree(d->d_ms);
    if (d->h_status) (void)hipHostFree(d->h_status);
    if (d->h_offs) (void)hipHostFree(d->h_offs);
    *d = lacx_decoder{};
}
// lacx_decode (no handle): one decoder per device for the life of the process (never freed: releasing device memory from
// a static or thread-local destructor would race the HIP runtime's own shutdown), calls on one device take turns
struct SharedDecoder {
    std::mutex mu;
    lacx_decoder dec;
};
constexpr int kMaxDecodeDevices = 64;
std::mutex g_shared_mu;
SharedDecoder* g_shared[kMaxDecodeDevices] = {};
}  // namespace

int lacx_decoder_create(int device, lacx_decoder** out) {
    if (!out) return LACX_E_INVALID;
    lacx_decoder* d = new lacx_decoder();
    d->device = device;
    *out = d;
    return LACX_OK;
}

void lacx_decoder_destroy(lacx_decoder* d) {
    if (!d) return;
    decoder_release(d);
    delete d;
}

int lacx_decoder_decode(lacx_decoder* d, const uint8_t* lac, uint64_t size, int32_t* left, int32_t* right, uint64_t frames,
                        float* device_ms) {
    if (!d) return decode_fail(LACX_E_INVALID, "null decoder");
    lacx_stream_info info;
    const int prc = lacx_stream_parse(lac, size, &info);
    if (prc) return prc;
    if (!left || (info.channels == 2 && !right)) return decode_fail(LACX_E_INVALID, "output arrays missing");
    if (frames != info.frames) return decode_fail(LACX_E_INVALID, "output arrays do not match the stream's frame count");
    if (device_ms) *device_ms = 0.f;
    if (lacx_device_count() <= 0) return decode_fail(LACX_E_DEVICE, "no usable HIP device");
#define DEC_TRY(call, what)                                                                                  \
    do {                                                                                                     \
        const hipError_t _e = (call);                                                                        \
        if (_e != hipSuccess) {                                                                              \
            rc = decode_fail(LACX_E_DEVICE, std::string(what) + ": " + hipGetErrorString(_e));               \
            goto done;                                                                                       \
        }                                                                                                    \
    } while (0)
    int rc = LACX_OK;
    const uint32_t nb = info.blocks;
    const bool v2 = info.version == 2;
    const uint64_t entry = v2 ? 4u : 8u;
    const uint64_t head = 14 + entry * nb, pay = size - head;
    const size_t noffs = 2 * ((size_t)nb + 1);  // byte offsets, then frame offsets
    int prev_device = -1;  // the caller's current device is put back on the way out
    if (!d->ready) {
        if (d->device < 0) DEC_TRY(hipGetDevice(&d->device), "hipGetDevice");
    }
    DEC_TRY(hipGetDevice(&prev_device), "hipGetDevice");
    if (prev_device == d->device) prev_device = -1;
    else DEC_TRY(hipSetDevice(d->device), "hipSetDevice");
    if (!d->ready) {
        DEC_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking), "hipStreamCreate");
        DEC_TRY(hipEventCreate(&d->e0), "hipEventCreate");
        DEC_TRY(hipEventCreate(&d->e1), "hipEventCreate");
        d->ready = true;
    }
    if (pay + kDecodeTailPad > d->pay_cap) {
        if (d->d_pay) (void)hipFree(d->d_pay);
        d->d_pay = nullptr;
        d->pay_cap = 0;
        const uint64_t cap = pay + pay / 8 + kDecodeTailPad;
        DEC_TRY(hipMalloc((void**)&d->d_pay, cap), "hipMalloc(payload)");
        d->pay_cap = cap;
    }
    if (nb > d->blocks_cap) {
        if (d->d_offs) (void)hipFree(d->d_offs);
        if (d->d_status) (void)hipFree(d->d_status);
        if (d->d_ms) (void)hipFree(d->d_ms);
        if (d->h_status) (void)hipHostFree(d->h_status);
        if (d->h_offs) (void)hipHostFree(d->h_offs);
        d->d_offs = nullptr;
        d->d_status = nullptr;
        d->d_ms = nullptr;
        d->h_status = nullptr;
        d->h_offs = nullptr;
        d->blocks_cap = 0;
        const uint32_t cap = nb + nb / 8 + 16;
        DEC_TRY(hipMalloc((void**)&d->d_offs, 2 * ((size_t)cap + 1) * sizeof(unsigned long long)), "hipMalloc(offsets)");
        DEC_TRY(hipMalloc((void**)&d->d_status, (size_t)cap * sizeof(uint32_t)), "hipMalloc(status)");
        DEC_TRY(hipMalloc((void**)&d->d_ms, cap), "hipMalloc(flags)");
        DEC_TRY(hipHostMalloc((void**)&d->h_status, (size_t)cap * sizeof(uint32_t), 0), "hipHostMalloc(status)");
        DEC_TRY(hipHostMalloc((void**)&d->h_offs, 2 * ((size_t)cap + 1) * sizeof(unsigned long long), 0), "hipHostMalloc(offsets)");
        d->blocks_cap = cap;
    }
    if (frames > d->pcm_cap || (info.channels == 2 && !d->d_right)) {
        if (d->d_left) (void)hipFree(d->d_left);
        if (d->d_right) (void)hipFree(d->d_right);
        d->d_left = d->d_right = nullptr;
        d->pcm_cap = 0;
        DEC_TRY(hipMalloc((void**)&d->d_left, frames * sizeof(int32_t)), "hipMalloc(left)");
        DEC_TRY(hipMalloc((void**)&d->d_right, frames * sizeof(int32_t)), "hipMalloc(right)");
        d->pcm_cap = frames;
    }
    {
        unsigned long long* byte_off = d->h_offs;
        unsigned long long* frame_off = d->h_offs + nb + 1;
        byte_off[0] = frame_off[0] = 0;
        for (uint32_t b = 0; b < nb; ++b) {
            frame_off[b + 1] = frame_off[b] + be32(lac + 14 + entry * b);
            byte_off[b + 1] = v2 ? 0 : byte_off[b] + be32(lac + 18 + 8ull * b);
        }
        hipStream_t st = d->stream;
        DEC_TRY(hipMemsetAsync(d->d_pay + pay, 0, kDecodeTailPad, st), "memset");  // the bit reader's look-ahead (decode.hip)
        DEC_TRY(hipMemcpyAsync(d->d_offs, d->h_offs, noffs * sizeof(unsigned long long), hipMemcpyHostToDevice, st), "H2D offsets");
        DEC_TRY(hipMemcpyAsync(d->d_pay, lac + head, pay, hipMemcpyHostToDevice, st), "H2D payload");
        DEC_TRY(hipEventRecord(d->e0, st), "event record");
        int32_t* dr = info.channels == 2 ? d->d_right : nullptr;
        if (v2)
            DEC_TRY(launch_decode_serial(nb, info.channels, info.stereo_mode, info.bit_depth, d->d_pay, (uint32_t)(8ull * pay), d->d_offs + nb + 1,
                                         d->d_left, dr, d->d_status, d->d_ms, st), "decode launch");
        else
            DEC_TRY(launch_decode(nb, info.channels, info.stereo_mode, info.bit_depth, d->d_pay, d->d_offs, d->d_offs + nb + 1, d->d_left, dr,
                                  d->d_status, d->d_ms, st), "decode launch");
        DEC_TRY(hipEventRecord(d->e1, st), "event record");
        DEC_TRY(hipMemcpyAsync(d->h_status, d->d_status, (size_t)nb * sizeof(uint32_t), hipMemcpyDeviceToHost, st), "D2H status");
        DEC_TRY(hipStreamSynchronize(st), "synchronize");
        if (device_ms) (void)hipEventElapsedTime(device_ms, d->e0, d->e1);
        for (uint32_t b = 0; b < nb; ++b) {
            if (d->h_status[b]) {  // the first failing block, like the reference's message (lac/decoder.cpp:24-32)
                static const char* const kWhat[] = {"", "block header", "channel header", "residual", "padding", "sample overflow",
                                                    "trailing bytes", "sample outside the bit depth", "not reached", "residual beyond 2^30"};
                rc = decode_fail(LACX_E_RUNTIME, "[decode-error] block=" + std::to_string(b) + " " +
                                                     (d->h_status[b] < 10 ? kWhat[d->h_status[b]] : "?"));
                goto done;
            }
        }
        // the two channels leave on two streams' worth of copy engine time: issue both, then wait
        DEC_TRY(hipMemcpyAsync(left, d->d_left, frames * sizeof(int32_t), hipMemcpyDeviceToHost, st), "D2H left");
        if (info.channels == 2) DEC_TRY(hipMemcpyAsync(right, d->d_right, frames * sizeof(int32_t), hipMemcpyDeviceToHost, st), "D2H right");
        DEC_TRY(hipStreamSynchronize(st), "synchronize");
    }
done:
#undef DEC_TRY
    if (prev_device >= 0) (void)hipSetDevice(prev_device);
    return rc;
}

int lacx_decode(int device, const uint8_t* lac, uint64_t size, int32_t* left, int32_t* right, uint64_t frames,
                float* device_ms) {
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = -1;
    if (dev < 0 || dev >= kMaxDecodeDevices) {
        if (lacx_device_count() <= 0) {  // (parse errors come first, as before)
            lacx_stream_info info;
            const int prc = lacx_stream_parse(lac, size, &info);
            return prc ? prc : decode_fail(LACX_E_DEVICE, "no usable HIP device");
        }
        return decode_fail(LACX_E_DEVICE, "HIP device ordinal out of range");
    }
    SharedDecoder* sd = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_shared_mu);
        if (!g_shared[dev]) {
            g_shared[dev] = new SharedDecoder();
            g_shared[dev]->dec.device = dev;
        }
        sd = g_shared[dev];
    }
    std::lock_guard<std::mutex> lock(sd->mu);
    lacx_decoder* d = &sd->dec;
    return lacx_decoder_decode(d, lac, size, left, right, frames, device_ms);
}

}  // extern "C"
