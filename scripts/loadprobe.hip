// Diagnostic micro-benchmark (not part of the product): latency of a workgroup's first 64 KB of global loads
// (4 x 16 B per thread, 1024 threads) at the start of a workgroup, with and without a long ALU phase after it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(1024) void k_probe(const uint4* __restrict__ in, uint32_t* out, unsigned long long* acc, int spin0, int jitter, int pair, int heavy, int tail) {
    extern __shared__ uint32_t lds[];
    if (threadIdx.x == 0) lds[0] = 1;
    int spin = spin0;
    if (jitter) spin = (int)(spin0 * (0.5f + (float)((blockIdx.x * 2654435761u) >> 24) / 256.0f));
    if (heavy) __syncthreads();  // all 16 waves issue their loads at the same moment
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned g = blockIdx.x / 16u, r = blockIdx.x % 16u;
    const unsigned region = pair ? (g * 8u + (r & 7u)) : blockIdx.x;  // pair: workgroups w and w+8 read the same 64 KB
    const uint4* p = in + (size_t)region * 4096 + threadIdx.x * 4;
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    uint32_t x = a.x ^ b.y ^ c.z ^ d.w;
    asm volatile("" : "+v"(x));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (heavy) {  // power-hungry: eight independent 64-bit multiply-add chains
        unsigned long long a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
        for (int i = 0; i < spin / 8; ++i) {
            a0 = a0 * 6364136223846793005ull + 1; a1 = a1 * 6364136223846793005ull + 3; a2 = a2 * 6364136223846793005ull + 5;
            a3 = a3 * 6364136223846793005ull + 7; a4 = a4 * 6364136223846793005ull + 9; a5 = a5 * 6364136223846793005ull + 11;
            a6 = a6 * 6364136223846793005ull + 13; a7 = a7 * 6364136223846793005ull + 15;
        }
        x = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
    } else {
        for (int i = 0; i < spin; ++i) x = x * 1664525u + 1013904223u;
    }
    out[(size_t)blockIdx.x * 1024 + threadIdx.x] = x;
    if (tail && threadIdx.x == 0) {  // a plan record written by one lane, byte by byte
        volatile unsigned char* pb = reinterpret_cast<volatile unsigned char*>(out) + (size_t)blockIdx.x * 4096 + 8192u * 1024u;
        for (int i = 0; i < 300; ++i) pb[i] = (unsigned char)(x + i);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&acc[0], t1 - t0);
        atomicAdd(&acc[1], 1ull);
    }
}

int main() {
    const int nwg = 3516;
    uint4* in; uint32_t* out; unsigned long long* acc;
    CHECK(hipMalloc(&in, (size_t)nwg * 65536)); CHECK(hipMalloc(&out, (size_t)nwg * 4096 * 2 + (8192u * 1024u))); CHECK(hipMalloc(&acc, 16));
    CHECK(hipMemset(in, 1, (size_t)nwg * 65536));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    for (int cfg = 0; cfg < 4; ++cfg) {
        const int spin = 400000;
        const int jitter = cfg & 1;
        const int pair = 0;
        const int heavy = cfg >> 1;
        const int tail = 0;
        const int ldsb = 120 * 1024;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(acc, 0, 16));
            hipLaunchKernelGGL(k_probe, dim3(nwg), dim3(1024), ldsb, 0, in, out, acc, spin, jitter, pair, heavy, tail);
            CHECK(hipDeviceSynchronize());
            unsigned long long h[2]; CHECK(hipMemcpy(h, acc, 16, hipMemcpyDeviceToHost));
            if (rep == 2) printf("tail %d heavy %d pair %d spin %6d jitter %d lds %6d: first-load latency %.0f shader cycles per wave (avg over %llu waves)\n", tail, heavy, pair, spin, jitter, ldsb, (double)h[0] / h[1], h[1]);
        }
    }
    return 0;
}
