// emit_device.h -- the device emit of one channel block (walks of emit_core.h + block scans + tile copy-out), shared by
// the emit fused into the whole-block analysis kernel (k_analyze.hip) and the stand-alone / repair kernel k_emit
// (k_emit.hip).  Device code only.
#pragma once
#include "device_util.h"
#include "emit_core.h"

namespace lacx {

// suffix-min scan of tabNX (first non-zero index per chunk) -> exclusive: min over later chunks
template <class G, class M>
__device__ __forceinline__ int32_t scan_nx_part1(M& sh, int tid, int32_t* wtot) {
    const int lane = tid & 63, wave = tid >> 6;
    int32_t v = sh.tabNX[tid];
    int32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t o = __shfl_down(inc, d, 64);
        if (lane + d < 64) inc = o < inc ? o : inc;
    }
    if (lane == 0) wtot[wave] = inc;
    return inc;
}

template <class G, class M>
__device__ __forceinline__ void scan_nx_part2(M& sh, int tid, int32_t inc, const int32_t* wtot, int32_t n) {
    constexpr int NW = G::T / 64;
    const int lane = tid & 63, wave = tid >> 6;
    int32_t later = n;
    for (int w = wave + 1; w < NW; ++w) later = wtot[w] < later ? wtot[w] : later;
    int32_t next = __shfl_down(inc, 1, 64);
    if (lane == 63) next = n;
    sh.tabNX[tid] = next < later ? next : later;
}


#ifndef LACX_WALK2_SINK
#define LACX_WALK2_SINK 0  // 1: the register-accumulator form (A/B)
#endif
// Walk 2 for the common shape of a channel block (device only): the whole bitstream fits ONE output tile (every 16-bit
// block, most 24-bit ones), the lane's chunk is complete and lies inside one partition, and that partition is coded with
// plain Rice tokens (mode 0 adaptive / 3 static) -- all of it for every lane of the wave, so the choice is one scalar
// branch.  Against the general walk (emit_walk_t, which stays for everything else) this one carries no token grammar
// selects, no zero-run look-ahead loads, no 64-bit word index and no tile bounds checks: per sample a shift, a mask, one
// 64-bit funnel insert and, every other sample or so, one LDS word store.  A thread's tokens are one contiguous bit range:
// its first and last word may hold a neighbour's bits (ds_or), the words in between are its own (plain store into the
// zeroed tile).  `pos` = bit position of the thread's first token in the channel block (< 2^22).
template <class G, class M>
__device__ __forceinline__ void emit_walk2_rice(const Thread<G>& th, M& sh, uint32_t* __restrict__ words, uint32_t pos,
                                                bool is_static, uint32_t k0) {
    const int t = th.tid;
#if LACX_WALK2_SINK
    uint64_t acc = 0;              // pending bits, left-aligned
    uint32_t fill = pos & 31u;     // number of pending bits (< 32 between samples)
    uint32_t word = pos >> 5;
    bool shared = true;            // the first word may also hold the previous thread's bits
    auto put = [&](uint32_t value, uint32_t n) {  // n in 1..32, value < 2^n, MSB first
        acc |= (uint64_t)value << (64u - fill - n);
        fill += n;
        if (fill >= 32u) {
            const uint32_t w = (uint32_t)(acc >> 32);
            if (shared) {
                if (w) atomicOr(&words[word], w);
            } else {
                words[word] = w;
            }
            acc <<= 32;
            fill -= 32u;
            ++word;
            shared = false;
        }
    };
#else
    // Every token goes straight to its place: OR-ed into the one or two words it covers (the tile is zeroed; tokens of
    // different threads meet in a word at most at the ends of their ranges, and a ds_or is as cheap as a store).  No
    // carried accumulator, no data-dependent flush: the only chain from sample to sample is the bit position.
    auto put = [&](uint32_t value, uint32_t n) {  // n in 1..32, value < 2^n, MSB first
        const uint32_t sh_ = pos & 31u;
        const uint64_t v64 = (uint64_t)value << (64u - sh_ - n);
        const uint32_t hi = (uint32_t)(v64 >> 32), lo = (uint32_t)v64;
        uint32_t* w = &words[pos >> 5];
        if (hi) atomicOr(w, hi);
        if (lo) atomicOr(w + 1, lo);
        pos += n;
    };
#endif
#pragma unroll 4
    for (int i = 0; i < G::CH; ++i) {
        const uint32_t u = sh.u[i * G::T + t] & 0x3FFFFFFFu;
        const uint32_t k = is_static ? k0 : (uint32_t)sh.xp.o.kin[i * G::T + t];
        uint32_t q = u >> k;
        const uint32_t rem = u & ((1u << k) - 1u);  // k <= 31
        if (q + k <= 30u) {  // the whole token in one insert of at most 31 bits
            put((((1u << q) - 1u) << (k + 1u)) | rem, q + 1u + k);
        } else {  // a long unary part (rare): 32 ones at a time, then the rest and the stop bit + remainder
            while (q >= 32u) {
                put(0xFFFFFFFFu, 32u);
                q -= 32u;
            }
            if (q) put((1u << q) - 1u, q);
            put(rem, k + 1u);
        }
    }
#if LACX_WALK2_SINK
    if (fill) atomicOr(&words[word], (uint32_t)(acc >> 32));  // the last, partial word is shared with the next thread
#endif
}

// Emit of one channel block from the residual in sh.u (plain zigzag values, block scans of tabP / tabNZ / tabNX done,
// plan fields loaded): walk 1 (Rice parameter per sample + token bits), bit offsets, walk 2 into 48 KiB LDS tiles,
// copy-out.  `resolve` is called once by all threads (it may contain barriers) before the first byte leaves the
// workgroup and yields the address the channel block's bitstream goes to; false from it abandons the emit.  Returns
// true when the whole bitstream was written.  Shared by k_emit and the emit fused into the analysis kernel.
template <class G, class M, class Resolve>
__device__ __forceinline__ bool emit_body(M& sh, Thread<G>& th, uint32_t n, uint8_t* __restrict__ out,
                                          uint32_t* __restrict__ err_flag, Resolve&& resolve, const int tid,
                                          const bool ablate_stores, const uint32_t slot_bytes STAMP_PARAMS) {
    (void)out;
    const bool narrow = sh.tabP[G::T] < kNarrowLimit;
    const bool adaptive0 = sh.p == 0 && (sh.part_mode_k[0] >> 5) != 3;  // stateful Rice::adapt_k walk
    if (adaptive0) {
        if (narrow) {
            phase_a<G, true>(th, sh);
        } else {
            phase_a<G, false>(th, sh);
        }
        __syncthreads();
    }
    STAMP(25);
    auto orw = [](uint32_t* w, uint32_t v) { atomicOr(w, v); };
    auto stw = [](uint32_t* w, uint32_t v) { *w = v; };
    // walk 1: Rice parameter per sample + token bits of the chunk
    const unsigned long long mybits = narrow ? emit_walk<G, true>(th, sh, nullptr, 0, orw, stw)
                                             : emit_walk<G, false>(th, sh, nullptr, 0, orw, stw);
    STAMP(26);
    __syncthreads();  // every thread is done with the sample prefix sums: tabP becomes the bit-offset table
    sh.tabP[tid] = mybits;
    {
        // sum scan of the bit counts (tabP only)
        const int lane = tid & 63, wave = tid >> 6;
        const unsigned long long inc = wave_scan_add_u64(mybits);
        if (lane == 63) sh.wtotP[wave] = inc;
        __syncthreads();
        unsigned long long base = 0;
        for (int w = 0; w < wave; ++w) base += sh.wtotP[w];
        sh.tabP[tid] = base + inc - mybits;
        if (tid == G::T - 1) sh.tabP[G::T] = base + inc;
        __syncthreads();
    }
    const unsigned long long total_bits = sh.tabP[G::T] + sh.header_bits;
    const unsigned long long nbytes = (total_bits + 7u) >> 3;
    if (tid == 0 && (nbytes != sh.payload_bytes || sh.err)) atomicOr(err_flag, 1u);
    if (nbytes != sh.payload_bytes || sh.err) return false;  // uniform: never write outside the planned byte range
    const unsigned long long mypos = sh.tabP[tid] + sh.header_bits;
    STAMP(27);

    // walk 2: tokens into 48 KiB LDS tiles, copied out tile by tile
    uint8_t* base = nullptr;
    for (unsigned long long bit0 = 0; bit0 < nbytes * 8u; bit0 += (unsigned long long)kEmitTileWords * 32u) {
        if (bit0 != 0) lds_barrier();  // every thread has copied its part of the previous tile out of LDS
        {
            // only the words this tile's bytes occupy (+ what the 16-byte copy-out may read past them)
            const unsigned long long left_bytes = nbytes - (bit0 >> 3);
            const int zw = left_bytes >= (unsigned long long)kEmitTileWords * 4u ? kEmitTileWords
                                                                                 : (int)((((uint32_t)left_bytes + 15u) >> 4) * 4u + 8u);
            const int zero_words = zw < kEmitTileWords ? zw : kEmitTileWords;
            for (int i = tid; i < zero_words; i += G::T) sh.xp.o.obits[i] = 0;
        }
        lds_barrier();
        STAMP(28);
        BitTile tile{sh.xp.o.obits, bit0, (uint32_t)kEmitTileWords};
        if (bit0 == 0) emit_header(th, sh, &tile, orw);
        const unsigned long long tile_end = bit0 + (unsigned long long)kEmitTileWords * 32u;
        // the lean walk where the whole bitstream is in this tile and the wave's chunks are plain Rice (see emit_walk2_rice)
        bool lean = false;
        uint32_t lean_k0 = 0;
        bool lean_static = false;
        if (bit0 == 0 && nbytes * 8u <= (unsigned long long)kEmitTileWords * 32u) {  // uniform
            const uint32_t p = sh.p, parts = sh.parts;
            const uint32_t pbase = p ? (n >> p) : n;
            uint32_t part = p ? ((uint32_t)th.a / pbase) : 0u;
            if (part >= parts) part = parts - 1u;
            const uint32_t pend = (part + 1u == parts) ? n : (part + 1u) * pbase;
            const uint32_t mk = sh.part_mode_k[part];
            const uint32_t mode = mk >> 5;
            lean_k0 = mk & 31u;
            lean_static = mode == 3u;
            lean = wave_all(th.cnt == G::CH && (uint32_t)th.a + (uint32_t)G::CH <= pend && (mode == 0u || mode == 3u) && lean_k0 <= 30u);
        }
        if (lean) {
            emit_walk2_rice<G>(th, sh, sh.xp.o.obits, (uint32_t)mypos, lean_static, lean_k0);
        } else if (mypos < tile_end && mypos + mybits > bit0) {
            if (narrow) {
                emit_walk<G, true>(th, sh, &tile, mypos, orw, stw);
            } else {
                emit_walk<G, false>(th, sh, &tile, mypos, orw, stw);
            }
        }
        STAMP(29);
        lds_barrier();
        STAMP(30);
        if (bit0 == 0 && !resolve(&base)) return false;  // uniform
        const unsigned long long byte0 = bit0 >> 3;
        const unsigned long long left = nbytes - byte0;
        const uint32_t count = left < (unsigned long long)kEmitTileWords * 4u ? (uint32_t)left : (uint32_t)kEmitTileWords * 4u;
        // Copy-out in 16-byte stores on 16-byte boundaries of the destination (which may be pinned host
        // memory behind PCIe: whole, aligned segments matter there); the unaligned head and tail go bytewise.
        uint8_t* dst = base + byte0;
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u);
        const uint32_t head = mis ? (16u - mis < count ? 16u - mis : count) : 0u;
        const uint32_t nvec = (count - head) >> 4;
        const uint32_t* tw = sh.xp.o.obits;
        auto tile_byte = [&](uint32_t i) { return (uint8_t)(tw[i >> 2] >> (24u - 8u * (i & 3u))); };
        if (ablate_stores) continue;  // timing ablation only (LACX_DEBUG_SKIP bit 11)
        if (slot_bytes) {
            // Staging slot (16-byte aligned, padded): whole 16-byte vectors only, stored write-through (sc1) so that
            // the hand-off to the streaming packer needs no release fence (cdna_hip_programming.md, Guideline 16, R1).
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)slot_bytes, 0x00020000);
            const uint32_t nv = (count + 15u) >> 4;
            for (uint32_t v = tid; v < nv; v += G::T) {
                u32x4 o;
                o.x = __builtin_bswap32(tw[4u * v]);
                o.y = __builtin_bswap32(tw[4u * v + 1u]);
                o.z = __builtin_bswap32(tw[4u * v + 2u]);
                o.w = __builtin_bswap32(tw[4u * v + 3u]);
                __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, (int)((uint32_t)byte0 + 16u * v), 0, 16 /* sc1 */);
            }
            continue;
        }
        if ((uint32_t)tid < head) dst[tid] = tile_byte((uint32_t)tid);
        {
            const uint32_t r = head & 3u, j0 = head >> 2;
            uint4* vdst = reinterpret_cast<uint4*>(dst + head);
            for (uint32_t v = tid; v < nvec; v += G::T) {
                const uint32_t j = j0 + 4u * v;
                uint32_t w[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const uint32_t jj = j + q < (uint32_t)kEmitTileWords ? j + q : (uint32_t)kEmitTileWords - 1u;
                    w[q] = __builtin_bswap32(tw[jj]);  // bytes of the stream in memory order
                }
                uint4 o;
                o.x = __builtin_amdgcn_alignbyte(w[1], w[0], r);
                o.y = __builtin_amdgcn_alignbyte(w[2], w[1], r);
                o.z = __builtin_amdgcn_alignbyte(w[3], w[2], r);
                o.w = __builtin_amdgcn_alignbyte(w[4], w[3], r);
                vdst[v] = o;
            }
        }
        {
            const uint32_t t0 = head + (nvec << 4);
            if (t0 + (uint32_t)tid < count) dst[t0 + tid] = tile_byte(t0 + (uint32_t)tid);
        }
        // no barrier and no wait for the stores here: the workgroup may retire while they are still on their way
        STAMP(31);
    }
    return true;
}

}  // namespace lacx
