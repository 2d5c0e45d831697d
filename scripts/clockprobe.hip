// Diagnostic micro-benchmark (not part of the product): shader clock under load and the issue cost of the
// integer instructions the analysis kernel is built from, at 4 waves per SIMD (1024-thread blocks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(1024) void k_op(uint32_t* out, uint64_t* clk, int iters, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = b + 77u, d = a + 3u;
    uint64_t A = ((uint64_t)a << 20) | b, B = ((uint64_t)c << 12) | d;
    double fa = 0.0, fb = 1.0, fx = (double)(int)(a & 0xFFFF), fy = (double)(int)(b & 0xFFFF), fz = (double)(int)(c & 0xFF);
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (OP == 0) { a = a + b; b = b ^ a; c = c + d; d = d ^ c; }                 // 4 independent-ish 32-bit ops
            if (OP == 1) { A = A + B; B = B + (A >> 3); }                                 // 64-bit add + shift
            if (OP == 2) { A = (uint64_t)a * b + B; a = (uint32_t)A ^ c; B += 1; }        // v_mad_u64_u32
            if (OP == 3) { a = __umulhi(a, b) + c; b += 0x10001u; }                        // v_mul_hi_u32
            if (OP == 4) { a += (A < B) ? 1u : 2u; A += 0x12345u; B += 0x54321u; }          // 64-bit compare
            if (OP == 5) { a = (uint32_t)__shfl_up((int)a, 1, 64) + b; }                   // ds_bpermute shuffle
            if (OP == 6) { a = a + (uint32_t)__popcll(__ballot((a >> (k & 15)) & 1u)); }   // ballot + popcount
            if (OP == 7) { a = (uint32_t)__clz((int)(a | 1u)) + b; b = b * 3u + 1u; }       // ffbh + mul_lo
            if (OP == 8) { a = (a >> (b & 31)) + c; b = b + 1u; }                            // variable shift
            if (OP == 9) { A = (uint64_t)((int64_t)(int32_t)a * (int64_t)(int32_t)b) + A; B = (uint64_t)((int64_t)(int32_t)c * (int64_t)(int32_t)d) + B; }  // 2 x v_mad_i64_i32, independent chains
            if (OP == 10) { fa = __builtin_fma(fx, fy, fa); fb = __builtin_fma(fy, fz, fb); }  // 2 x v_fma_f64
            if (OP == 11) { a = (uint32_t)((int32_t)a * (int32_t)b) + c; d = (uint32_t)((int32_t)d * (int32_t)b) + c; }  // 2 x v_mul_lo + add (mad_u32)
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (uint32_t)A + (uint32_t)B + (uint32_t)(long long)(fa + fb);
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int OP>
int run(const char* name, int ops_per_k) {
    uint32_t* out; uint64_t* clk;
    CHECK(hipMalloc(&out, 256 * 1024 * 4)); CHECK(hipMalloc(&clk, 16));
    const int iters = 4000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_op<OP>, dim3(256), dim3(1024), 0, 0, out, clk, iters, 123u);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    uint64_t h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double shader_ghz = (double)h[0] / ((double)h[1] * 10.0);  // memrealtime ticks at 100 MHz
    const double instr = (double)iters * 16 * ops_per_k;              // per wave
    // 16 waves per CU = 4 per SIMD; cycles per wave-instruction per SIMD
    const double cyc_per_instr_simd = (ms * 1e-3 * shader_ghz * 1e9) / (instr * 4.0);
    printf("%-28s %8.3f ms  shader %.2f GHz  %.2f cycles / wave-instr / SIMD (4 waves per SIMD)\n", name, ms, shader_ghz, cyc_per_instr_simd);
    hipFree(out); hipFree(clk);
    return 0;
}

int main() {
    run<0>("v_add/v_xor u32", 4);
    run<1>("u64 add + shift", 3);
    run<2>("v_mad_u64_u32 (+2)", 3);
    run<3>("v_mul_hi_u32 (+2)", 3);
    run<4>("u64 compare (+4)", 5);
    run<5>("shfl_up (ds_bpermute) (+1)", 2);
    run<6>("ballot+popc (+3)", 5);
    run<7>("ffbh + mul_lo (+3)", 4);
    run<8>("variable shift (+3)", 4);
    run<9>("2 x v_mad_i64_i32", 2);
    run<10>("2 x v_fma_f64", 2);
    run<11>("2 x mul_lo_u32 + add", 4);
    return 0;
}
