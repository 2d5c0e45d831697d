// Reads like the reference's own tests (tests/test_e2e.cpp:888-930, tests/test_predictors.cpp:47-69):
// the C++ mirror classes must compile with the reference's call syntax and raise its exception types.
// Exit 0: all checks passed; exit 77: no HIP device (only the argument checks ran).
#include <cstdio>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "codec/block/encoder.hpp"
#include "codec/lac/decoder.hpp"
#include "codec/lac/encoder.hpp"

// a caller's header struct with the reference's field names (src/codec/frame/frame_header.hpp:7-14)
struct FrameHeader {
    uint16_t sync = 0;
    uint8_t version = 0, channels = 0, stereo_mode = 0;
    uint32_t sample_rate = 0;
    uint8_t bit_depth = 0, reserved = 0;
};

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

template <class Ex, class F>
static bool throws(F&& f) {
    try { f(); } catch (const Ex&) { return true; } catch (...) { return false; }
    return false;
}

// LAC::Encoder over several devices (ref src/codec/lac/encoder.cpp:385-465 spreads the blocks over its workers itself):
// the default (every visible device), explicit lists with repeated ordinals, set_thread_count untouched -- same bytes.
static int fanout_checks() {
    const size_t frames = 16384 * 9 + 4321;
    std::vector<int32_t> left(frames), right(frames);
    uint32_t s = 12345;
    int32_t a = 0, b = 0;
    for (size_t i = 0; i < frames; ++i) {
        s = s * 1664525u + 1013904223u;
        a += (int32_t)((s >> 20) & 0xFF) - 128;
        b += (int32_t)((s >> 8) & 0x7F) - 64;
        if (a > 30000 || a < -30000) a = 0;
        if (b > 30000 || b < -30000) b = 0;
        left[i] = a;
        right[i] = (i / 40000) % 2 ? b : a / 2;
    }
    LAC::Encoder one(12, 2, 48000, 16);
    one.set_devices({0});
    const std::vector<uint8_t> want = one.encode(left, right);
    CHECK(want.size() > 22);
    LAC::Encoder all(12, 2, 48000, 16);  // default: every visible device
    CHECK(all.encode(left, right) == want);
    for (const std::vector<int>& list : {std::vector<int>{0, 0}, std::vector<int>{0, 0, 0, 0}, std::vector<int>{0, 0, 0, 0, 0, 0, 0}}) {
        LAC::Encoder e(12, 2, 48000, 16);
        e.set_devices(list, 1);
        LAC::ThreadCollector tc;
        CHECK(e.encode(left, right, &tc) == want);
        CHECK(e.encode(left, right) == want);  // and again on the same handle
        LAC::Encoder copy = e;                 // a copy keeps the list
        CHECK(copy.encode(left, right) == want);
    }
    const int n = lacx_device_count();
    if (n > 1) {
        std::vector<int> list;
        for (int i = 0; i < n; ++i) list.push_back(i);
        LAC::Encoder e(12, 2, 48000, 16);
        e.set_devices(list, 1);
        CHECK(e.encode(left, right) == want);
    }
    {   // errors keep the reference's wording and the stream-wide index
        std::vector<int32_t> l2 = left;
        l2[16384 * 7 + 3] = 40000;
        LAC::Encoder e(12, 2, 48000, 16);
        e.set_devices({0, 0, 0}, 1);
        bool ok = false;
        try { e.encode(l2, right); } catch (const std::invalid_argument& ex) {
            ok = std::string(ex.what()) == "left sample at index " + std::to_string(16384 * 7 + 3) + " is outside the configured PCM bit depth";
        }
        CHECK(ok);
    }
    LAC::Decoder dec;
    std::vector<int32_t> ol, orr;
    dec.decode(want.data(), want.size(), ol, orr);
    CHECK(ol == left && orr == right);
    std::printf("mirror api (fan-out): %s\n", fails ? "FAILED" : "ok");
    return fails ? 1 : 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "fanout") return fanout_checks();
    std::vector<int32_t> left(3000), right(3000);
    for (size_t i = 0; i < left.size(); ++i) { left[i] = (int32_t)((i * 37) % 2000) - 1000; right[i] = left[i] / 2; }
    // argument validation (ref src/codec/lac/encoder.cpp:220-237)
    CHECK(throws<std::invalid_argument>([&] { LAC::Encoder e(12, 2, 48000, 16); e.encode({}, {}); }));
    CHECK(throws<std::invalid_argument>([&] { LAC::Encoder e(12, 2, 48000, 16); std::vector<int32_t> r2(10); e.encode(left, r2); }));
    CHECK(throws<std::invalid_argument>([&] { LAC::Encoder e(12, 2, 22050, 16); e.encode(left, right); }));
    CHECK(throws<std::invalid_argument>([&] { LAC::Encoder e(12, 2, 48000, 20); e.encode(left, right); }));
    CHECK(throws<std::invalid_argument>([&] { LAC::Encoder e(12, 3, 48000, 16); e.encode(left, right); }));
    // decoder: malformed input is a std::runtime_error before any device is needed (ref tests/test_e2e.cpp:497-545)
    {
        LAC::Decoder dec;
        std::vector<int32_t> ol, orr;
        std::vector<uint8_t> junk(10, 0);
        CHECK(throws<std::runtime_error>([&] { dec.decode(junk.data(), junk.size(), ol, orr); }));
        CHECK(throws<std::runtime_error>([&] { dec.decode(nullptr, 0, ol, orr, nullptr); }));
    }
    if (lacx_device_count() < 1) {
        CHECK(throws<std::runtime_error>([&] { LAC::Encoder e(12, 2, 48000, 16); e.encode(left, right); }));
        std::printf("mirror api: argument checks %s (no HIP device)\n", fails ? "FAILED" : "ok");
        return fails ? 1 : 77;
    }
    LAC::Encoder enc(12, 2, 48000, 16);
    enc.set_zero_run_enabled(true);
    enc.set_partitioning_enabled(true);
    enc.set_thread_count(2);
    LAC::ThreadCollector tc;
    const std::vector<uint8_t> bytes = enc.encode(left, right, &tc);
    CHECK(bytes.size() > 22 && bytes[0] == 0x4C && bytes[1] == 0x41 && bytes[2] == 3);
    CHECK(tc.snapshot().size() == 1);
    {   // and back, the way the reference's round-trip helper does it (tests/test_e2e.cpp:110-125)
        LAC::ThreadCollector dtc;
        LAC::Decoder dec(&dtc);
        dec.set_thread_count(2);
        std::vector<int32_t> ol, orr;
        FrameHeader hdr;
        dec.decode(bytes.data(), bytes.size(), ol, orr, &hdr);
        CHECK(ol == left && orr == right);
        CHECK(hdr.sync == 0x4C41 && hdr.version == 3 && hdr.channels == 2 && hdr.stereo_mode == 2 && hdr.sample_rate == 48000 &&
              hdr.bit_depth == 16 && hdr.reserved == 0);
        CHECK(dtc.snapshot().size() == 1);
        std::vector<uint8_t> cut(bytes.begin(), bytes.end() - 2);
        CHECK(throws<std::runtime_error>([&] { dec.decode(cut.data(), cut.size(), ol, orr); }));
        CHECK(ol.empty() && orr.empty());
        LAC::Encoder mono(12, 0, 44100, 24);
        const std::vector<uint8_t> mb = mono.encode(left, {});
        dec.decode(mb.data(), mb.size(), ol, orr, nullptr);
        CHECK(ol == left && orr.empty());
    }
    Block::Encoder benc(12);
    benc.set_zero_run_enabled(false);
    benc.set_partitioning_enabled(false);
    std::vector<int32_t> ramp(128);
    for (size_t i = 0; i < ramp.size(); ++i) ramp[i] = (int32_t)i * 3;
    const std::vector<uint8_t> blk = benc.encode(ramp);  // ramp -> fixed predictor (ref tests/test_predictors.cpp:53-57)
    CHECK(!blk.empty() && blk[0] == 0);
    {
        std::vector<int32_t> l2 = left;
        l2[5] = 40000;
        CHECK(throws<std::invalid_argument>([&] { LAC::Encoder e(12, 2, 48000, 16); e.encode(l2, right); }));
    }
    std::printf("mirror api: %s\n", fails ? "FAILED" : "ok");
    return fails ? 1 : 0;
}
