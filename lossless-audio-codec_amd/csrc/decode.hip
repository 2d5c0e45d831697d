// decode.hip -- CDNA4 (gfx950) kernels of the LAC v3 decoder (the product's own check that a .lac it
// produced gives back the PCM, on a box where the reference is absent).
//
// What the format allows to run in parallel is the block: inside a block the two channel bitstreams follow each other
// byte-aligned but without a length field, every token's length depends on the Rice parameter, and the Rice parameter
// depends on every sample decoded before it (ref src/codec/block/decoder.cpp:64-520, src/codec/rice/rice.hpp:45-114).
// So: ONE LANE PER BLOCK, both channels one after the other, all blocks of the stream at once -- the duration is one
// block's serial chain whatever the stream's length (up to the chip's ~65 000 resident lanes = 18 h of audio), and the
// throughput comes from the number of blocks.  Per-lane state that must be indexed lives in LDS, one column per lane:
// the last 256 residual magnitudes of the stateful Rice adaptation and the predictor's history.
//   k_decode      bitstream -> residuals -> samples (fixed / FIR / LPC synthesis), planar int32, per-block status
//   k_ms_inverse  mid/side -> left/right where the block's flag says so, and the bit-depth range check
//                                                                        (ref src/codec/lac/decoder.cpp:48-65,30-46)
// The adaptive Rice parameter uses the encoder's division-free formulation (kmean / biased_k of analyze_core.h, proven
// against Rice::adapt_k there); it assumes zigzag residuals below 2^30 like the encoder does, and a stream with a larger
// one is refused (status 9) rather than decoded differently from the reference.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "analyze_core.h"
#include "kernels.h"

namespace lacx {

namespace {

constexpr int kDecThreads = 64;
constexpr uint32_t kModeBin = 2, kModeStatic = 3;  // (0 = adaptive Rice, 1 = zero-run)  ref block/constants.hpp
constexpr uint32_t kZeroRunMin = 4, kZeroRunK = 2;

// Per-lane state that must be indexed, in LDS, one column per ACTIVE lane (row stride = active lanes of the wave, so a
// wave that decodes one block needs 1.3 KiB and many such waves share a CU).
struct DecMem {
    uint32_t* ring_;  // [256][cols] the last 256 residual magnitudes (stateful adaptation's drift window)
    int32_t* hist_;   // [32][cols]  the last 32 reconstructed samples (only LPC orders above 12 read it)
    int16_t* coef_;   // [32][cols]  the channel block's Q15 coefficients
    uint32_t cols;
    __device__ __forceinline__ uint32_t& ring(uint32_t slot, int lane) { return ring_[slot * cols + (uint32_t)lane]; }
    __device__ __forceinline__ int32_t& hist(uint32_t slot, int lane) { return hist_[slot * cols + (uint32_t)lane]; }
    __device__ __forceinline__ int16_t& coef(uint32_t slot, int lane) { return coef_[slot * cols + (uint32_t)lane]; }
};
constexpr size_t kDecBytesPerCol = 256 * 4 + 32 * 4 + 32 * 2;

// MSB-first bit reader over a byte stream in global memory (ref src/codec/bitstream/bit_reader.hpp).  A lane's stream is
// latency-bound -- every token's position depends on the one before -- so the reader keeps the next bits in a 64-bit
// register (buf: `have` valid bits from r.pos on, left-aligned, zeros below) and fetches the stream as 32-bit words one
// word AHEAD of the one it appends (nxt, byte-swapped only when it is appended, so that nothing waits for the load
// before it is needed); one word per top-up keeps the top-up at eight instructions.  Positions are 32-bit, relative to
// the block (a block's bitstream is far below 2^32 bits).
// Bounds are not checked read by read: a read past the end of the block yields bits of the next block or of the
// kDecodeTailPad zero bytes the host appends to the payload, and the caller compares r.pos with r.nbits once per trip
// (overrun()); every loop whose length the stream controls (the coefficient list, a long unary run, the partition
// table) checks BEFORE it reads.  Worst overshoot of one trip from r.pos <= r.nbits: 2 tag bits + 64 unary bits at hand
// + 32 remainder bits = 13 bytes, plus the reader's 64 buffered bits and one 4-byte word of look-ahead: 25 bytes < kDecodeTailPad.
struct BitIn {
    const uint8_t* p;
    uint32_t nbits, pos, have, widx;  // widx: index of the 32-bit word held (still raw) in nxt = the word of bit pos + have
    unsigned long long buf;
    uint32_t nxt;
};

__device__ __forceinline__ uint32_t load_word(const uint8_t* p, uint32_t idx) {
    uint32_t w;
    __builtin_memcpy(&w, p + 4ull * idx, 4);
    return w;
}
__device__ __forceinline__ void reader_seek(BitIn& r, uint32_t bitpos) {
    const uint32_t o = bitpos & 31u;
    r.pos = bitpos;
    r.widx = (bitpos >> 5) + 1u;
    r.buf = ((unsigned long long)__builtin_bswap32(load_word(r.p, r.widx - 1u)) << 32) << o;  // the word's bits from bitpos on
    r.have = 32u - o;
    r.nxt = load_word(r.p, r.widx);
}
__device__ __forceinline__ void reader_init(BitIn& r, const uint8_t* p, uint32_t nbits) {
    r.p = p;
    r.nbits = nbits;
    reader_seek(r, 0);
}
__device__ __forceinline__ bool overrun(const BitIn& r) { return r.pos > r.nbits; }
// one step of topping the register up (have <= 32): the next 32-bit word of the stream goes in behind the bits at hand,
// the word after it is fetched (the low bits of buf beyond `have` are zero: invariant)
__device__ __forceinline__ void refill_step(BitIn& r) {
    r.buf |= (unsigned long long)__builtin_bswap32(r.nxt) << (32u - r.have);
    r.have += 32u;
    ++r.widx;
    r.nxt = load_word(r.p, r.widx);
}
// afterwards the register holds at least 33 valid bits (a second step only when it was empty)
__device__ __forceinline__ void refill(BitIn& r) {
    if (r.have <= 32u) {
        refill_step(r);
        if (r.have <= 32u) refill_step(r);
    }
}
__device__ __forceinline__ void consume(BitIn& r, uint32_t n) {  // n <= have <= 64
    r.buf = (r.buf << (n >> 1)) << (n - (n >> 1));  // two shifts: n may be 64
    r.have -= n;
    r.pos += n;
}
// n <= 32 bits that are already in the register
__device__ __forceinline__ uint32_t take(BitIn& r, uint32_t n) {
    const uint32_t v = n ? (uint32_t)(r.buf >> (64u - n)) : 0u;
    consume(r, n);
    return v;
}
__device__ __forceinline__ uint32_t get_bits(BitIn& r, uint32_t n) {  // n <= 32
    refill(r);
    return take(r, n);
}
// unary: ones terminated by a zero; more than max_q ones is a malformed stream (ref block/decoder.cpp:76-86).
// The long form: the run of ones goes on beyond the bits at hand -- register by register, with an eye on the end of the
// block.  (The per-sample loop handles the common case, a terminator among the bits at hand, inline.)
__device__ bool get_unary_slow(BitIn& r, uint32_t max_q, uint32_t& q) {
    unsigned long long c = 0;
    for (;;) {
        c += r.have;
        consume(r, r.have);
        if (c > (unsigned long long)max_q || r.pos >= r.nbits) return false;
        refill(r);
        const unsigned long long iv = ~r.buf;
        const uint32_t ones = iv ? (uint32_t)__clzll((long long)iv) : 64u;
        if (ones < r.have) {
            c += ones;
            consume(r, ones + 1u);
            break;
        }
    }
    q = (uint32_t)c;
    return c <= (unsigned long long)max_q;
}
// Expects a refilled register.
__device__ __forceinline__ bool get_unary(BitIn& r, uint32_t max_q, uint32_t& q) {
    const unsigned long long inv = ~r.buf;  // the invalid low bits of buf are zero, so they read as terminators
    const uint32_t ones = inv ? (uint32_t)__clzll((long long)inv) : 64u;
    if (ones < r.have) {
        consume(r, ones + 1u);
        q = ones;
        return ones <= max_q;
    }
    return get_unary_slow(r, max_q, q);
}
__device__ __forceinline__ bool get_rice(BitIn& r, uint32_t k, uint32_t& value) {  // k <= 31
    refill(r);
    uint32_t q = 0;
    if (!get_unary(r, 0xFFFFFFFFu >> k, q)) return false;
    if (r.have < k) refill(r);
    value = (q << k) | take(r, k);
    return true;
}
__device__ __forceinline__ int32_t unzigzag(uint32_t u) {
    return (u & 1u) ? (int32_t)(-(long long)((u >> 1) + 1u)) : (int32_t)(u >> 1);
}
__device__ __forceinline__ uint32_t zigzag(int32_t v) { return ((uint32_t)v << 1) ^ (uint32_t)(v >> 31); }

// Rice::AdaptState in the encoder's feed-forward form: prefix sum, count, the sum of the last 256 magnitudes (ring in
// LDS) and the two flag counts over the last 96 samples (flags in two 96-bit shift registers).
struct Adapt {
    unsigned long long sum, wsum;
    uint32_t count, large, zero;
    uint32_t lf[3], zf[3];
};
__device__ __forceinline__ void adapt_reset(Adapt& a) {
    a.sum = a.wsum = 0;
    a.count = a.large = a.zero = 0;
    a.lf[0] = a.lf[1] = a.lf[2] = a.zf[0] = a.zf[1] = a.zf[2] = 0;
}
// One more sample of magnitude u (if `on`); returns the parameter for the next one (ref rice.hpp:45-114 /
// encoder.cpp:72-77), or `k` unchanged when `on` is false.  Straight-line for stateless partitions.
__device__ __forceinline__ uint32_t adapt_next(Adapt& a, uint32_t u, bool on, uint32_t k, bool stateless, DecMem& dm, int lane) {
    a.sum += on ? u : 0u;
    a.count += on ? 1u : 0u;
    const uint32_t cnt = a.count ? a.count : 1u;
    // 32-bit form while the sum allows it (it does for every block of ordinary material)
    const uint32_t km = (a.sum >> 31) == 0ull ? kmean32((uint32_t)a.sum, cnt) : kmean(a.sum, cnt);
    uint32_t kn = km > 31u ? 31u : km;
    if (!stateless && on) {
        // drift window: the last 256 magnitudes
        const uint32_t slot = (a.count - 1u) & 255u;
        if (a.count > 256u) a.wsum -= dm.ring(slot, lane);
        dm.ring(slot, lane) = u;
        a.wsum += u;
        // micro window: flags of the last 96 samples
        const uint32_t q = km >= 31u ? 0u : (u >> km);
        const uint32_t fl = q > 3u ? 1u : 0u, fz = q == 0u ? 1u : 0u;
        a.large += fl - (a.lf[2] >> 31);
        a.zero += fz - (a.zf[2] >> 31);
        a.lf[2] = (a.lf[2] << 1) | (a.lf[1] >> 31);
        a.lf[1] = (a.lf[1] << 1) | (a.lf[0] >> 31);
        a.lf[0] = (a.lf[0] << 1) | fl;
        a.zf[2] = (a.zf[2] << 1) | (a.zf[1] >> 31);
        a.zf[1] = (a.zf[1] << 1) | (a.zf[0] >> 31);
        a.zf[0] = (a.zf[0] << 1) | fz;
        kn = biased_k<false>(km, a.sum, a.sum - a.wsum, a.large | (a.zero << 16), a.count);
    }
    return on ? kn : k;
}

// One channel block: header, partition table, residual tokens, synthesis, zero padding to the byte
// (ref block/decoder.cpp:64-520).  The 64 lanes of a wave decode 64 different blocks, so everything per sample is ONE loop
// that every lane walks in step -- one sample per trip whatever the partition, its mode (data, not control flow: the
// four token grammars are alternatives inside the trip), a zero run in progress (its zeros come out one per trip) or
// the predictor (the synthesis of sample i follows its residual at once: it only needs earlier samples).  Written as
// four loops per partition and a synthesis pass per predictor type, lanes in different loops would take turns.
// 0 = ok, else a status code.
__device__ uint32_t decode_channel_block(BitIn& r, uint32_t n, int32_t* __restrict__ out, DecMem& dm, int lane) {
    const uint32_t type = get_bits(r, 8);
    const int order = (int)get_bits(r, 8);
    if (overrun(r) || type > 2u) return 2;
    if (type == 2u) {
        if (order <= 0 || order > 32 || (uint32_t)order >= n) return 2;
    } else if (type == 1u) {
        if (order != 2) return 2;
    } else if (order > 4) {
        return 2;
    }
    if (type == 2u) {
        // the coefficient list is as long as the stream says (up to 32 x 16 bits): it must lie inside the block before
        // a single bit of it is fetched (a block that ends right behind a type-2 header must not be read past its pad)
        if (r.pos + 16u * (uint32_t)order > r.nbits) return 2;
        for (int i = 0; i < order; ++i) dm.coef((uint32_t)i, lane) = (int16_t)get_bits(r, 16);
        for (int i = order; i < 12; ++i) dm.coef((uint32_t)i, lane) = 0;  // the synthesis always walks twelve taps
        if (overrun(r)) return 2;
    }
    const uint32_t control = get_bits(r, 8);
    if (overrun(r) || (control & 0x10u)) return 2;
    const bool pflag = (control & 0x80u) != 0u;
    const uint32_t p = control & 0x0Fu, cmode = (control >> 5) & 3u;
    if ((pflag && p == 0u) || (!pflag && p != 0u) || p > (uint32_t)kMaxPartitionOrder) return 2;
    if (p > 0u && (n >> p) < (uint32_t)kMinPartition) return 2;
    const uint32_t parts = (p == 0u || (n >> p) == 0u) ? 1u : (1u << p);
    const uint32_t base = parts == 1u ? n : (n >> p);
    const uint32_t table_pos = r.pos;  // (mode:2, k:5) per partition, read when the partition starts
    if (r.pos + 7u * parts > r.nbits) return 2;
    reader_seek(r, r.pos + 7u * parts);
    const bool stateless = p > 0u;

    Adapt a;
    adapt_reset(a);
    uint32_t mode = 0, k = 0, seg_end = 0, part = 0, zeros_left = 0, st = 0;
    // Every predictor as twelve taps on the last twelve samples, a shift, and a number of warm-up samples that are taken
    // as they are: fixed orders 1..4 with their binomial taps and no shift, the FIR predictor (3 x1 - x2) >> 2 after two
    // samples, LPC with its Q15 coefficients (zero beyond the order) from the first sample on -- the window starts as
    // zeros, which is what "taps that reach before the block start are left out" amounts to.  Window and taps live in
    // registers; a tap is one multiply-add.
    int32_t hw[12], cw[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) {
        hw[t] = 0;
        cw[t] = (type == 2u) ? (int32_t)dm.coef((uint32_t)t, lane) : 0;
    }
    if (type == 1u) {
        cw[0] = 3;
        cw[1] = -1;
    } else if (type == 0u) {
        cw[0] = order;                                              // 1 2 3 4
        cw[1] = order == 2 ? -1 : (order == 3 ? -3 : (order == 4 ? -6 : 0));
        cw[2] = order == 3 ? 1 : (order == 4 ? 4 : 0);
        cw[3] = order == 4 ? -1 : 0;
    }
    const uint32_t pshift = type == 2u ? 15u : (type == 1u ? 2u : 0u);
    const uint32_t warm = type == 2u ? 0u : (type == 1u ? 2u : (uint32_t)order);
    for (uint32_t i = 0; i < n; ++i) {
        if (i == seg_end) {  // a partition starts
            BitIn t = r;
            reader_seek(t, table_pos + 7u * part);
            mode = get_bits(t, 2);
            k = get_bits(t, 5);
            if (part == 0u && mode != cmode) {
                st = 2;
                break;
            }
            seg_end += (part + 1u == parts) ? n - base * (parts - 1u) : base;
            ++part;
            adapt_reset(a);
        }
        // The plain trip: every lane of the wave that is still decoding sits in a partition whose tokens are bare Rice
        // codes -- static Rice, or adaptive Rice of a partitioned block (stateless: prefix mean only) -- no run in
        // progress, at most twelve taps, and the unary part ends among the bits at hand.  Most of a music stream is that
        // (this encoder picks static Rice for nearly every partition of ordinary material), and the trip then is a third
        // of the general one below: no tag, no mode selects, no windows.  Wave-uniform choice per trip.
        {
            const bool plain = zeros_left == 0u && (mode == kModeStatic || (mode == 0u && stateless)) && !(type == 2u && order > 12);
            bool lean = __ballot(!plain) == 0ull;
            uint32_t ones = 0;
            if (lean) {
                refill(r);
                const unsigned long long inv = ~r.buf;  // the invalid low bits of buf are zero: they read as terminators
                ones = inv ? (uint32_t)__clzll((long long)inv) : 64u;
                lean = __ballot(ones >= r.have) == 0ull;
            }
            if (lean) {
                consume(r, ones + 1u);
                uint32_t bad = ones > (0xFFFFFFFFu >> k) ? 3u : 0u;
                if (r.have < k) refill(r);
                const uint32_t u = (ones << k) | take(r, k);
                if (overrun(r)) bad = 3u;
                if (!bad && (u >> 30)) bad = 9u;
                long long acc = 0;
#pragma unroll
                for (int t = 0; t < 12; ++t) acc += (long long)cw[t] * (long long)hw[t];
                const long long s = (long long)unzigzag(u) + (i >= warm ? (acc >> pshift) : 0ll);
                if ((long long)(int32_t)s != s && !bad) bad = 5u;
                if (bad) {
                    st = bad;
                    break;
                }
                if (mode == 0u) {  // stateless adaptation: the prefix mean of the partition (ref block/encoder.cpp:72-77)
                    a.sum += u;
                    a.count += 1u;
                    const uint32_t km = (a.sum >> 31) == 0ull ? kmean32((uint32_t)a.sum, a.count) : kmean(a.sum, a.count);
                    k = km > 31u ? 31u : km;
                }
                out[i] = (int32_t)s;
#pragma unroll
                for (int t = 11; t > 0; --t) hw[t] = hw[t - 1];
                hw[0] = (int32_t)s;
                continue;
            }
        }
        // One token, whatever the grammar: [2-bit tag] [unary quotient] [remainder / sign / 32-bit escape], each part
        // present or not, chosen by selects -- the trip has the same few branches for every mode (refills, the long
        // unary form, the stateful adaptation, the error exit).  A zero run in progress yields its zeros one per trip.
        const bool in_run = zeros_left != 0u;
        zeros_left -= in_run ? 1u : 0u;
        const bool is_bin = mode == kModeBin, is_zr = mode == 1u;
        const bool tagged = !in_run && (is_bin || is_zr);
        refill(r);
        const uint32_t tag = tagged ? (uint32_t)(r.buf >> 62) : 0u;
        consume(r, tagged ? 2u : 0u);
        const bool run_token = tagged && is_zr && tag == 1u;
        const bool has_unary = !in_run && (!tagged || (is_bin ? tag == 3u : tag <= 1u));
        const uint32_t kk = run_token ? kZeroRunK : k;
        uint32_t bad = (tagged && is_zr && tag == 3u) ? 3u : 0u;
        uint32_t q = 0;
        {
            const unsigned long long inv = ~r.buf;  // the invalid low bits of buf are zero: they read as terminators
            const uint32_t ones = inv ? (uint32_t)__clzll((long long)inv) : 64u;
            if (has_unary && ones >= r.have) {  // the run of ones goes on beyond the bits at hand (rare)
                if (!get_unary_slow(r, 0xFFFFFFFFu >> kk, q)) bad = 3u;
            } else {
                q = has_unary ? ones : 0u;
                consume(r, has_unary ? ones + 1u : 0u);
            }
            if (q > (0xFFFFFFFFu >> kk)) bad = 3u;
        }
        const uint32_t rem_bits =
            in_run ? 0u : (has_unary ? kk : (is_bin ? ((tag == 1u || tag == 2u) ? 1u : 0u) : ((is_zr && tag == 2u) ? 32u : 0u)));
        if (r.have < rem_bits) refill(r);
        const uint32_t rem = take(r, rem_bits);
        const uint32_t value = has_unary ? ((q << kk) | rem) : rem;
        const bool small_bin = tagged && is_bin && (tag == 1u || tag == 2u);  // +-1, +-2: tag and sign bit
        const uint32_t u = small_bin ? zigzag(rem ? -(int32_t)tag : (int32_t)tag) : ((run_token || in_run) ? 0u : value);
        bool adapt = mode != kModeStatic;  // a static partition keeps the k of its table entry
        if (in_run) adapt = !stateless;    // stateful streams adapt on every zero, stateless ones did it at the token
        if (run_token) {
            const unsigned long long run = (unsigned long long)value + kZeroRunMin;
            if (run > (unsigned long long)(seg_end - i)) bad = 3u;
            zeros_left = (uint32_t)run - 1u;
            if (stateless) {  // the count jumps by the run, the parameter is recomputed once
                a.count += (uint32_t)run;
                const uint32_t km = kmean(a.sum, a.count);
                k = km > 31u ? 31u : km;
                adapt = false;
            }
        }
        if (overrun(r)) bad = 3u;
        if (!bad && (u >> 30)) bad = 9u;
        // synthesis: twelve taps (orders above 12 -- valid streams, none from this encoder -- add theirs from LDS)
        long long acc = 0;
#pragma unroll
        for (int t = 0; t < 12; ++t) acc += (long long)cw[t] * (long long)hw[t];
        if (type == 2u && order > 12) {
            const int taps = order < (int)i ? order : (int)i;
            for (int t = 13; t <= taps; ++t)
                acc += (long long)dm.coef((uint32_t)t - 1u, lane) * (long long)dm.hist((i - (uint32_t)t) & 31u, lane);
        }
        const long long s = (long long)unzigzag(u) + (i >= warm ? (acc >> pshift) : 0ll);
        if ((long long)(int32_t)s != s && !bad) bad = 5u;  // the reference rejects a sample that leaves int32
        if (bad) {
            st = bad;
            break;
        }
        k = adapt_next(a, u, adapt, k, stateless, dm, lane);
        out[i] = (int32_t)s;
        if (type == 2u && order > 12) dm.hist(i & 31u, lane) = (int32_t)s;
#pragma unroll
        for (int t = 11; t > 0; --t) hw[t] = hw[t - 1];
        hw[0] = (int32_t)s;
    }
    if (st) return st;
    while (r.pos & 7u) {  // zero padding to the byte (ref bit_reader.hpp consume_zero_padding_to_byte)
        if (get_bits(r, 1) || overrun(r)) return 4;
    }
    return 0;
}

}  // namespace

// lanes_per_wave (a power of two, 1..64): how many blocks one wave decodes -- fewer blocks per wave mean more waves to
// interleave on a SIMD while the stream has few enough blocks that the idle lanes do not matter (the launcher picks).
__global__ __launch_bounds__(kDecThreads) void k_decode(uint32_t num_blocks, int channels, int stereo_mode,
                                                        uint32_t lanes_per_wave,
                                                        const uint8_t* __restrict__ payload,
                                                        const unsigned long long* __restrict__ byte_off,
                                                        const unsigned long long* __restrict__ frame_off,
                                                        int32_t* __restrict__ left, int32_t* __restrict__ right,
                                                        uint32_t* __restrict__ status, uint8_t* __restrict__ ms_flag) {
    extern __shared__ __align__(16) unsigned char dec_raw[];
    DecMem dm;
    dm.cols = lanes_per_wave;
    dm.ring_ = reinterpret_cast<uint32_t*>(dec_raw);
    dm.hist_ = reinterpret_cast<int32_t*>(dec_raw + (size_t)256 * 4 * lanes_per_wave);
    dm.coef_ = reinterpret_cast<int16_t*>(dec_raw + (size_t)(256 + 32) * 4 * lanes_per_wave);
    const int lane = (int)threadIdx.x;
    if ((uint32_t)lane >= lanes_per_wave) return;
    const uint32_t blk = blockIdx.x * lanes_per_wave + threadIdx.x;
    if (blk >= num_blocks) return;
    const uint32_t n = (uint32_t)(frame_off[blk + 1] - frame_off[blk]);
    BitIn r;
    reader_init(r, payload + byte_off[blk], (uint32_t)(8ull * (byte_off[blk + 1] - byte_off[blk])));
    uint32_t st = 0;
    uint32_t ms = stereo_mode == 1 ? 1u : 0u;
    if (n == 0u || n > (uint32_t)kMaxBlock) st = 1;
    if (!st && channels == 2 && stereo_mode == 2) {  // per-block flag byte (ref lac/decoder.cpp)
        const uint32_t flag = get_bits(r, 8);
        if (overrun(r) || flag > 1u) st = 1;
        ms = flag;
    }
    if (!st) st = decode_channel_block(r, n, left + frame_off[blk], dm, lane);
    if (!st && channels == 2) st = decode_channel_block(r, n, right + frame_off[blk], dm, lane);
    if (!st && r.pos != r.nbits) st = 6;  // trailing bytes in the block
    status[blk] = st;
    ms_flag[blk] = (uint8_t)ms;
}

// The legacy version-2 container carries no compressed block sizes (ref lac/decoder.cpp:209-219): block i starts where
// block i-1 ended, so ONE lane walks the whole stream.  Kept for completeness of the reader (the encoder has written
// version 3 only since); status[] is set for the blocks up to and including the first that fails.
__global__ __launch_bounds__(kDecThreads) void k_decode_serial(uint32_t num_blocks, int channels, int stereo_mode,
                                                               const uint8_t* __restrict__ payload, uint32_t payload_bits,
                                                               const unsigned long long* __restrict__ frame_off,
                                                               int32_t* __restrict__ left, int32_t* __restrict__ right,
                                                               uint32_t* __restrict__ status, uint8_t* __restrict__ ms_flag) {
    extern __shared__ __align__(16) unsigned char dec_raw[];
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    DecMem dm;
    dm.cols = 1;
    dm.ring_ = reinterpret_cast<uint32_t*>(dec_raw);
    dm.hist_ = reinterpret_cast<int32_t*>(dec_raw + 256 * 4);
    dm.coef_ = reinterpret_cast<int16_t*>(dec_raw + (256 + 32) * 4);
    BitIn r;
    reader_init(r, payload, payload_bits);
    for (uint32_t blk = 0; blk < num_blocks; ++blk) {
        const uint32_t n = (uint32_t)(frame_off[blk + 1] - frame_off[blk]);
        uint32_t st = 0, ms = stereo_mode == 1 ? 1u : 0u;
        if (n == 0u || n > (uint32_t)kMaxBlock) st = 1;
        if (!st && channels == 2 && stereo_mode == 2) {
            const uint32_t flag = get_bits(r, 8);
            if (overrun(r) || flag > 1u) st = 1;
            ms = flag;
        }
        if (!st) st = decode_channel_block(r, n, left + frame_off[blk], dm, 0);
        if (!st && channels == 2) st = decode_channel_block(r, n, right + frame_off[blk], dm, 0);
        if (!st && blk + 1u == num_blocks && r.pos != r.nbits) st = 6;  // trailing frame payload
        status[blk] = st;
        ms_flag[blk] = (uint8_t)ms;
        if (st) {
            for (uint32_t b = blk + 1u; b < num_blocks; ++b) status[b] = 8;  // not reached
            break;
        }
    }
}

// grid = (blocks, tiles): the samples of block blockIdx.x in tiles of 1024
__global__ __launch_bounds__(256) void k_ms_inverse(int channels, int bit_depth,
                                                    const unsigned long long* __restrict__ frame_off,
                                                    int32_t* __restrict__ left, int32_t* __restrict__ right,
                                                    const uint8_t* __restrict__ ms_flag, uint32_t* __restrict__ status) {
    const uint32_t blk = blockIdx.x, tile = blockIdx.y;
    if (status[blk]) return;  // (uniform) the block did not decode
    const unsigned long long f0 = frame_off[blk];
    const uint32_t n = (uint32_t)(frame_off[blk + 1] - f0);
    const bool ms = channels == 2 && ms_flag[blk] != 0;
    const long long lo = bit_depth == 16 ? -32768 : -0x800000, hi = bit_depth == 16 ? 32767 : 0x7FFFFF;
    bool bad = false;
    for (uint32_t i = tile * 1024u + threadIdx.x; i < n && i < (tile + 1u) * 1024u; i += 256u) {
        long long l = left[f0 + i], rr = channels == 2 ? right[f0 + i] : 0;
        if (ms) {  // ref lac/decoder.cpp:48-65
            const long long m = l, s = rr;
            l = m + ((s + (s & 1)) >> 1);
            rr = l - s;
            left[f0 + i] = (int32_t)l;
            right[f0 + i] = (int32_t)rr;
        }
        bad = bad || l < lo || l > hi || (channels == 2 && (rr < lo || rr > hi));
    }
    if (bad) atomicMax(&status[blk], 7u);
}

static void launch_ms_inverse(uint32_t num_blocks, int channels, int bit_depth, const unsigned long long* frame_off,
                              int32_t* left, int32_t* right, const uint8_t* ms_flag, uint32_t* status, hipStream_t stream) {
    hipLaunchKernelGGL(k_ms_inverse, dim3(num_blocks, kMaxBlock / 1024), dim3(256), 0, stream, channels, bit_depth, frame_off,
                       left, right, ms_flag, status);
}

hipError_t launch_decode(uint32_t num_blocks, int channels, int stereo_mode, int bit_depth, const uint8_t* payload,
                         const unsigned long long* byte_off, const unsigned long long* frame_off, int32_t* left,
                         int32_t* right, uint32_t* status, uint8_t* ms_flag, hipStream_t stream) {
    if (num_blocks == 0) return hipSuccess;
    static const uint32_t lanes = [] {  // blocks per wave (see k_decode); LACX_DECODE_LANES: tuning knob, read once
        if (const char* v = std::getenv("LACX_DECODE_LANES")) {  // 1, 2, 4, ... 64
            const int x = std::atoi(v);
            if (x >= 1 && x <= 64 && (x & (x - 1)) == 0) return (uint32_t)x;
        }
        return 64u;
    }();
    const size_t smem = kDecBytesPerCol * lanes;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(kDecBytesPerCol * 64));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_decode, dim3((num_blocks + lanes - 1) / lanes), dim3(kDecThreads), smem, stream, num_blocks, channels,
                       stereo_mode, lanes, payload, byte_off, frame_off, left, right, status, ms_flag);
    launch_ms_inverse(num_blocks, channels, bit_depth, frame_off, left, right, ms_flag, status, stream);
    return hipGetLastError();
}

hipError_t launch_decode_serial(uint32_t num_blocks, int channels, int stereo_mode, int bit_depth, const uint8_t* payload,
                                uint32_t payload_bits, const unsigned long long* frame_off, int32_t* left, int32_t* right,
                                uint32_t* status, uint8_t* ms_flag, hipStream_t stream) {
    if (num_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_decode_serial, dim3(1), dim3(kDecThreads), kDecBytesPerCol, stream, num_blocks, channels, stereo_mode,
                       payload, payload_bits, frame_off, left, right, status, ms_flag);
    launch_ms_inverse(num_blocks, channels, bit_depth, frame_off, left, right, ms_flag, status, stream);
    return hipGetLastError();
}

}  // namespace lacx
