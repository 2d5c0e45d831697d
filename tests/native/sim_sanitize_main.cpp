// Driver for the sanitizer build of the kernel-phase simulator (tests/test_native_units.py): runs the analysis
// and emit phases over a handful of block shapes and materials; AddressSanitizer / UBSan abort on any
// out-of-bounds LDS-image access, shift or overflow in the phase code the HIP kernels are built from.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "lacx_types.h"

extern "C" int sim_block_encode(const int32_t* x, uint32_t n, int zero_run, int partitioning, int force_wide, uint8_t* out,
                                uint32_t cap);
extern "C" int sim_block_plan(const int32_t* x, uint32_t n, int zero_run, int partitioning, int geo, int force_wide,
                              lacx::ChannelPlan* out);

int main() {
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&s]() {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return (uint32_t)(s >> 32);
    };
    const uint32_t sizes[] = {16384, 16383, 4097, 4096, 257, 256, 33, 2, 1};
    std::vector<uint8_t> out(16384 * 8 + 64);
    int runs = 0;
    for (int kind = 0; kind < 6; ++kind) {
        for (uint32_t n : sizes) {
            std::vector<int32_t> x(n);
            int32_t walk = 0;
            for (uint32_t i = 0; i < n; ++i) {
                const int32_t r = (int32_t)(rnd() % 65536u) - 32768;
                switch (kind) {
                    case 0: x[i] = r; break;                                       // white noise, 16 bit
                    case 1: x[i] = r * 256 + (int32_t)(rnd() & 255u); break;       // loud 24 bit: 64-bit paths
                    case 2: x[i] = 0; break;                                       // silence
                    case 3: x[i] = (rnd() % 97u == 0) ? (r >> 6) : 0; break;       // sparse: zero runs
                    case 4: walk += (r >> 10); x[i] = walk; break;                 // random walk
                    default: x[i] = (int32_t)(i & 3u) - 1; break;                  // +-1: bin mode
                }
            }
            for (int wide = 0; wide < 8; wide += (n > 1000 ? 3 : 1)) {
                if (sim_block_encode(x.data(), n, 1, 1, wide, out.data(), (uint32_t)out.size()) < 0) return 2;
                ++runs;
            }
            lacx::ChannelPlan plan;
            if (n <= 256 && sim_block_plan(x.data(), n, 1, 1, 1, 0, &plan) != 0) return 3;
        }
    }
    std::printf("sanitized simulator runs: %d\n", runs);
    return 0;
}
