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

int lacx_decode(int device, const uint8_t* lac, uint64_t size, int32_t* left, int32_t* right, uint64_t frames,
                float* device_ms) {
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
    std::vector<unsigned long long> offs(2 * ((size_t)nb + 1));  // byte offsets, then frame offsets
    unsigned long long* byte_off = offs.data();
    unsigned long long* frame_off = offs.data() + nb + 1;
    byte_off[0] = frame_off[0] = 0;
    for (uint32_t b = 0; b < nb; ++b) {
        frame_off[b + 1] = frame_off[b] + be32(lac + 14 + entry * b);
        byte_off[b + 1] = v2 ? 0 : byte_off[b] + be32(lac + 18 + 8ull * b);
    }
    uint8_t* d_pay = nullptr;
    unsigned long long* d_offs = nullptr;
    int32_t *d_left = nullptr, *d_right = nullptr;
    uint32_t* d_status = nullptr;
    uint8_t* d_ms = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<uint32_t> status(nb);
    int prev_device = -1;  // the caller's current device is put back on the way out
    if (device >= 0) {
        DEC_TRY(hipGetDevice(&prev_device), "hipGetDevice");
        if (prev_device == device) prev_device = -1;
        else DEC_TRY(hipSetDevice(device), "hipSetDevice");
    }
    DEC_TRY(hipMalloc((void**)&d_pay, pay + kDecodeTailPad), "hipMalloc(payload)");  // the bit reader's look-ahead (decode.hip)
    DEC_TRY(hipMemset(d_pay + pay, 0, kDecodeTailPad), "memset");
    DEC_TRY(hipMemcpy(d_pay, lac + head, pay, hipMemcpyHostToDevice), "H2D payload");
    DEC_TRY(hipMalloc((void**)&d_offs, offs.size() * sizeof(unsigned long long)), "hipMalloc(offsets)");
    DEC_TRY(hipMemcpy(d_offs, offs.data(), offs.size() * sizeof(unsigned long long), hipMemcpyHostToDevice), "H2D offsets");
    DEC_TRY(hipMalloc((void**)&d_left, frames * sizeof(int32_t)), "hipMalloc(left)");
    if (info.channels == 2) DEC_TRY(hipMalloc((void**)&d_right, frames * sizeof(int32_t)), "hipMalloc(right)");
    DEC_TRY(hipMalloc((void**)&d_status, (size_t)nb * sizeof(uint32_t)), "hipMalloc(status)");
    DEC_TRY(hipMalloc((void**)&d_ms, nb), "hipMalloc(flags)");
    DEC_TRY(hipEventCreate(&e0), "hipEventCreate");
    DEC_TRY(hipEventCreate(&e1), "hipEventCreate");
    DEC_TRY(hipEventRecord(e0, nullptr), "event record");
    if (v2)
        DEC_TRY(launch_decode_serial(nb, info.channels, info.stereo_mode, info.bit_depth, d_pay, (uint32_t)(8ull * pay), d_offs + nb + 1,
                                     d_left, d_right, d_status, d_ms, nullptr), "decode launch");
    else
        DEC_TRY(launch_decode(nb, info.channels, info.stereo_mode, info.bit_depth, d_pay, d_offs, d_offs + nb + 1, d_left, d_right,
                              d_status, d_ms, nullptr), "decode launch");
    DEC_TRY(hipEventRecord(e1, nullptr), "event record");
    DEC_TRY(hipMemcpy(status.data(), d_status, (size_t)nb * sizeof(uint32_t), hipMemcpyDeviceToHost), "D2H status");
    if (device_ms) (void)hipEventElapsedTime(device_ms, e0, e1);
    for (uint32_t b = 0; b < nb; ++b) {
        if (status[b]) {  // the first failing block, like the reference's message (lac/decoder.cpp:24-32)
            static const char* const kWhat[] = {"", "block header", "channel header", "residual", "padding", "sample overflow",
                                                "trailing bytes", "sample outside the bit depth", "not reached", "residual beyond 2^30"};
            rc = decode_fail(LACX_E_RUNTIME, "[decode-error] block=" + std::to_string(b) + " " +
                                                 (status[b] < 10 ? kWhat[status[b]] : "?"));
            goto done;
        }
    }
    DEC_TRY(hipMemcpy(left, d_left, frames * sizeof(int32_t), hipMemcpyDeviceToHost), "D2H left");
    if (info.channels == 2) DEC_TRY(hipMemcpy(right, d_right, frames * sizeof(int32_t), hipMemcpyDeviceToHost), "D2H right");
done:
#undef DEC_TRY
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d_pay) (void)hipFree(d_pay);
    if (d_offs) (void)hipFree(d_offs);
    if (d_left) (void)hipFree(d_left);
    if (d_right) (void)hipFree(d_right);
    if (d_status) (void)hipFree(d_status);
    if (d_ms) (void)hipFree(d_ms);
    if (prev_device >= 0) (void)hipSetDevice(prev_device);
    return rc;
}

}  // extern "C"
