// Reads like the reference's own tests (tests/test_e2e.cpp:888-930, tests/test_predictors.cpp:47-69):
// the C++ mirror classes must compile with the reference's call syntax and raise its exception types.
// Exit 0: all checks passed; exit 77: no HIP device (only the argument checks ran).
#include <cstdio>
#include <cstdint>
#include <stdexcept>
#include <vector>

#include "codec/block/encoder.hpp"
#include "codec/lac/encoder.hpp"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

template <class Ex, class F>
static bool throws(F&& f) {
    try { f(); } catch (const Ex&) { return true; } catch (...) { return false; }
    return false;
}

int main() {
    std::vector<int32_t> left(3000), right(3000);
    for (size_t i = 0; i < left.size(); ++i) { left[i] = (int32_t)((i * 37) % 2000) - 1000; right[i] = left[i] / 2; }
    // argument validation (ref src/codec/lac/encoder.cpp:220-237)
    CHECK(throws<std::invalid_argument>([&] { LAC::Encoder e(12, 2, 48000, 16); e.encode({}, {}); }));
    CHECK(throws<std::invalid_argument>([&] { LAC::Encoder e(12, 2, 48000, 16); std::vector<int32_t> r2(10); e.encode(left, r2); }));
    CHECK(throws<std::invalid_argument>([&] { LAC::Encoder e(12, 2, 22050, 16); e.encode(left, right); }));
    CHECK(throws<std::invalid_argument>([&] { LAC::Encoder e(12, 2, 48000, 20); e.encode(left, right); }));
    CHECK(throws<std::invalid_argument>([&] { LAC::Encoder e(12, 3, 48000, 16); e.encode(left, right); }));
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
