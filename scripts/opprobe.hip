// Diagnostic micro-benchmark (not part of the product): issue cost per SIMD of the integer instructions the analysis
// kernels are (or could be) built from, as cycles per wave-instruction at 4 waves per SIMD (1024-thread blocks, one per
// CU) and at 1 wave per SIMD (256-thread blocks).  Four independent dependency chains per op, so the figure is
// throughput, not latency.   hipcc -O3 --offload-arch=gfx950 scripts/opprobe.hip -o /tmp/opprobe && /tmp/opprobe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef short v2s __attribute__((ext_vector_type(2)));

template <int OP, int T>
__global__ __launch_bounds__(T) void k_op(uint32_t* out, uint64_t* clk, int iters, uint32_t seed) {
    uint32_t a[4], b[4];
    uint64_t A[4];
    for (int q = 0; q < 4; ++q) {
        a[q] = threadIdx.x * 2654435761u + seed + q * 977u;
        b[q] = (a[q] ^ 0x9E3779B9u) | 1u;
        A[q] = ((uint64_t)a[q] << 20) | b[q];
    }
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 1) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(A[q]) : "v"(a[q]), "v"(b[q]) : "vcc");
                if (OP == 2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 3) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 4) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(A[q]) : "v"(a[q]), "v"(b[q]) : "vcc");
                if (OP == 5) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 6) asm volatile("v_dot2_i32_i16 %0, %0, %1, %0" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 7) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 8) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 9) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(A[q]) : "v"(b[q]));
                if (OP == 10) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(A[q]) : "v"(A[(q + 1) & 3]));
                if (OP == 11) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a[q]));
                if (OP == 12) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 13) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 14) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 15) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[q]) : "v"(b[q]) : "vcc");
                if (OP == 16) asm volatile("v_bfe_u32 %0, %0, %1, 5" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 17) asm volatile("v_cmp_lt_u64 vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %2, vcc" : "+v"(A[q]), "+v"(A[(q + 1) & 3]), "+v"(a[q]) : : "vcc");
                if (OP == 18) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 19) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 20) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 21) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 22) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 23) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 24) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 25) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 26) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 27) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 28) asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 29) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 30) asm volatile("v_mov_b32 %0, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 31) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[q]) : "v"(b[q]) : "vcc");
                if (OP == 32) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : "+v"(a[q]) : "v"(b[q]) : "vcc");
                if (OP == 33) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 34) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 35) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 36) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[q]) : "v"(b[q]) : "vcc");
                if (OP == 37) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[q]) : "v"(b[q]) : "vcc");
                if (OP == 38) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 39) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 40) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 41) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 42) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 43) asm volatile("v_cmp_lt_u32_e64 s[10:11], %0, %1" : "+v"(a[q]) : "v"(b[q]) : "s10", "s11");
                if (OP == 44) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 45) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 46) asm volatile("v_add_u32 %0, 7, %0" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 47) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 48) asm volatile("v_bfe_i32 %0, %0, 3, 5" : "+v"(a[q]) : "v"(b[q]));
                if (OP == 49) asm volatile("v_sad_u32 %0, %0, %1, %0" : "+v"(a[q]) : "v"(b[q]));
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t s = 0;
    for (int q = 0; q < 4; ++q) s += a[q] + b[q] + (uint32_t)A[q] + (uint32_t)(A[q] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int OP, int T>
double run1(int per_op) {
    uint32_t* out; uint64_t* clk;
    if (hipMalloc(&out, 256 * 1024 * 4) != hipSuccess || hipMalloc(&clk, 16) != hipSuccess) return -1;
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_op<OP, T>), dim3(256), dim3(T), 0, 0, out, clk, iters, 123u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    uint64_t h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double shader_ghz = (double)h[0] / ((double)h[1] * 10.0);  // memrealtime ticks at 100 MHz
    const double instr = (double)iters * 8 * 4 * per_op;              // per wave
    const double waves_per_simd = T / 256.0;
    hipFree(out); hipFree(clk);
    return (ms * 1e-3 * shader_ghz * 1e9) / (instr * waves_per_simd);
}

template <int OP>
void run(const char* name, int per_op = 1) {
    const double c4 = run1<OP, 1024>(per_op), c1 = run1<OP, 256>(per_op);
    printf("%-34s %6.2f cycles/wave-instr/SIMD at 4 waves/SIMD   %6.2f at 1 wave/SIMD\n", name, c4, c1);
}

int main() {
    run<0>("v_add_u32");
    run<24>("v_and_b32");
    run<25>("v_xor_b32");
    run<26>("v_sub_u32");
    run<27>("v_lshlrev_b32");
    run<28>("v_ashrrev_i32");
    run<29>("v_min_u32");
    run<30>("v_mov_b32");
    run<31>("v_cndmask_b32 (vcc fixed)");
    run<32>("v_cmp_lt_u32 (VOPC)");
    run<33>("v_lshl_add_u32");
    run<34>("v_and_or_b32");
    run<35>("v_add_u32_e64 (VOP3 encoding)");
    run<36>("v_add_co_u32");
    run<37>("v_addc_co_u32");
    run<38>("v_fma_f32");
    run<39>("v_mul_i32_i24");
    run<40>("v_or_b32");
    run<41>("v_max_i32");
    run<42>("v_add_f32");
    run<43>("v_cmp_lt_u32_e64 -> sgpr pair");
    run<44>("v_subrev_u32");
    run<45>("v_lshrrev_b32 by constant");
    run<46>("v_add_u32 inline constant");
    run<47>("v_add_u32 literal constant");
    run<48>("v_bfe_i32 constants");
    run<49>("v_sad_u32");
    run<13>("v_add3_u32");
    run<12>("v_lshrrev_b32");
    run<16>("v_bfe_u32");
    run<11>("v_ffbh_u32");
    run<18>("v_med3_i32");
    run<14>("v_perm_b32");
    run<23>("v_alignbit_b32");
    run<20>("v_bcnt_u32_b32");
    run<21>("v_pk_add_u16");
    run<15>("v_cmp_lt_u32 + v_cndmask", 2);
    run<19>("v_mov_b32_dpp row_shr:1");
    run<2>("v_mul_u32_u24");
    run<3>("v_mul_hi_u32_u24");
    run<22>("v_mad_u32_u24");
    run<5>("v_mad_i32_i24");
    run<6>("v_dot2_i32_i16");
    run<7>("v_mul_lo_u32");
    run<8>("v_mul_hi_u32");
    run<1>("v_mad_u64_u32");
    run<4>("v_mad_i64_i32");
    run<9>("v_lshrrev_b64");
    run<10>("v_lshl_add_u64");
    run<17>("v_cmp_lt_u64 + v_addc_co_u32", 2);
    return 0;
}
