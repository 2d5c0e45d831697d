// emit_core.h -- per-thread phases of the device-side bit emit.
//
// One workgroup turns one analysed channel block into its exact bitstream
// (ref src/codec/block/encoder.cpp:554-838; token grammars :585-771; bit order bitstream/bit_writer.cpp):
//   1. residual of the chosen predictor (phase_r of analyze_core.h) + block scans;
//   2. walk 1: the Rice parameter in force for every sample -- the same feed-forward formulation the
//      analysis uses (stateful Rice::adapt_k when unpartitioned, adapt_k_stateless inside partitions,
//      constant for static-Rice partitions) -- stored as one byte per sample, and the thread's token bits;
//   3. block scan of the bit counts -> every thread's bit offset;
//   4. walk 2: tokens OR-ed MSB-first into an LDS tile (big-endian words), tile copied out.
// As in analyze_core.h these are plain per-thread functions, shared by the HIP kernels (emit_device.h) and the
// host simulator (tests/native/sim_analyze.cpp); cross-thread steps live in the drivers.
#pragma once
#include "analyze_core.h"

namespace lacx {

template <class G>
struct EmitMem {
    uint32_t u[G::MAXN + 4];  // zigzag residual (transposed), plain; +4: lookahead pad
    union XP {
        int32_t x[G::MAXN];  // staged samples until the residual is formed
        EmitOut<G> o;        // output tile + Rice parameter per sample (analyze_core.h)
    } xp;
    uint64_t tabP[G::T + 1];  // chunk sums -> exclusive prefix; later: bit counts -> bit offsets
    int32_t tabNZ[G::T + 1];  // last non-zero index -> exclusive prefix max
    int32_t tabNX[G::T + 1];  // first non-zero index -> exclusive suffix min (n if none)
    uint32_t tabF[G::T + 1];  // packed micro-window flag counts of every chunk (phase A)
    uint16_t tabZM[G::T];     // ... and the flags themselves, sample by sample (zero quotient / large quotient)
    uint16_t tabLM[G::T];
    uint64_t wtotP[16];
    int32_t wtotZ[16];
    uint32_t wtotF[16];
    LpcSet lpc;  // entry 0 carries the plan's coefficients so that phase_r can be reused
    uint8_t part_mode_k[kMaxParts];
    uint32_t ptype, order, p, parts, cand, header_bits, payload_bytes;
    uint32_t err;
};

// plan -> shared memory (thread 0) ---------------------------------------------------------------
template <class M>
LACX_HD void emit_load_plan(M& sh, const ChannelPlan& pl, int tid = 0, int nthreads = 1) {
    // cooperative: every element is written by exactly one of the `nthreads` callers (a single caller does it all)
    const uint32_t parts = pl.partition_order ? (1u << pl.partition_order) : 1u;
    const uint32_t order = pl.order;
    const bool lpc = pl.predictor_type == 2;
    for (uint32_t i = (uint32_t)tid; i < (uint32_t)kMaxParts; i += (uint32_t)nthreads)
        sh.part_mode_k[i] = (i < parts) ? pl.part_mode_k[i] : (uint8_t)0;
    for (uint32_t i = (uint32_t)tid; i < 5u * 13u; i += (uint32_t)nthreads) {
        const uint32_t ci = i / 13u, t = i % 13u;
        sh.lpc.coef[ci][t] = (lpc && ci == 0u && t >= 1u && t <= order && t <= 12u) ? pl.coef[t - 1u] : (int16_t)0;
    }
    if (tid == 0) {
        sh.ptype = pl.predictor_type;
        sh.order = pl.order;
        sh.p = pl.partition_order;
        sh.parts = parts;
        sh.payload_bytes = pl.payload_bytes;
        sh.cand = pl.predictor_type == 0 ? pl.order : (pl.predictor_type == 1 ? 5u : 6u);
        for (int ci = 0; ci < 5; ++ci) sh.lpc.used[ci] = 0;
        if (lpc) sh.lpc.used[0] = pl.order;
        sh.lpc.pad = 0;
        sh.header_bits = 16u + (lpc ? 16u * pl.order : 0u) + 8u + 7u * parts;
        sh.err = 0;
    }
}

// first non-zero index of the chunk (n if none); input of the suffix-min scan
template <class G, class M>
LACX_HD void emit_first_nonzero(const Thread<G>& th, M& sh) {
    int32_t first = (int32_t)th.n;
    for (int i = th.cnt - 1; i >= 0; --i) {
        if ((sh.u[i * G::T + th.tid] & 0x3FFFFFFFu) != 0) first = th.a + i;
    }
    sh.tabNX[th.tid] = first;
}

// Token of one sample.  Returns its bit length; when W is non-null also writes it at bit `pos`.
struct BitTile {
    uint32_t* words;     // tile storage
    uint64_t bit0;       // first bit of the tile in the channel-block bitstream
    uint32_t nwords;
};

template <class Or>
LACX_HD void put_bits(const BitTile* tile, uint64_t pos, uint32_t value, uint32_t n, Or&& or_word) {
    // n in 1..32, value < 2^n; MSB-first.  Pieces outside the tile are dropped (another pass owns them).
    const uint64_t v64 = (uint64_t)value << (64u - n - (uint32_t)(pos & 31u));
    const uint32_t hi = (uint32_t)(v64 >> 32), lo = (uint32_t)v64;
    const uint64_t w = pos >> 5;
    const uint64_t w0 = tile->bit0 >> 5;
    if (w >= w0 && w < w0 + tile->nwords && hi) or_word(&tile->words[w - w0], hi);
    if (lo && w + 1 >= w0 && w + 1 < w0 + tile->nwords) or_word(&tile->words[w + 1 - w0], lo);
}

template <class Or>
LACX_HD void put_ones(const BitTile* tile, uint64_t pos, uint32_t q, Or&& or_word) {
    while (q >= 32u) {
        put_bits(tile, pos, 0xFFFFFFFFu, 32u, or_word);
        pos += 32u;
        q -= 32u;
    }
    if (q) put_bits(tile, pos, (1u << q) - 1u, q, or_word);
}

// Bit writer of one thread in walk 2.  A thread's tokens form one contiguous bit range of the channel block, so
// it assembles them in a 64-bit register and writes whole 32-bit words: words strictly inside its range belong
// to it alone (plain store into the zeroed tile), only its first and its last word can hold a neighbour's bits
// and are OR-ed.  Words outside the tile are dropped (another pass owns them).
template <class Or, class St>
struct TokenSink {
    uint32_t* words;  // tile storage (copies of the BitTile fields: they must live in registers, not behind a pointer)
    uint64_t w0;      // first stream word of the tile
    uint32_t nwords;
    Or& or_word;
    St& st_word;
    uint64_t acc;   // pending bits, left-aligned
    uint32_t fill;  // number of pending bits (< 32 between calls)
    uint64_t word;  // stream word index the top 32 bits of acc belong to
    bool shared;    // that word may also hold bits of the previous thread

    LACX_HD TokenSink(BitTile t, uint64_t pos, Or& o, St& s)
        : words(t.words), w0(t.bit0 >> 5), nwords(t.nwords), or_word(o), st_word(s), acc(0),
          fill((uint32_t)(pos & 31u)), word(pos >> 5), shared(true) {}
    LACX_HD void flush_word(uint32_t w, bool with_or) {
        if (word < w0 || word >= w0 + nwords) return;
        if (with_or) {
            if (w) or_word(&words[word - w0], w);
        } else {
            st_word(&words[word - w0], w);
        }
    }
    LACX_HD void put(uint32_t value, uint32_t n) {  // n in 0..32, value < 2^n, MSB first
        if (n == 0u) return;
        acc |= (uint64_t)value << (64u - fill - n);
        fill += n;
        if (fill >= 32u) {
            flush_word((uint32_t)(acc >> 32), shared);
            acc <<= 32;
            fill -= 32u;
            ++word;
            shared = false;
        }
    }
    LACX_HD void put_rice(uint32_t u, uint32_t k) {  // q ones, a zero, k remainder bits
        uint32_t q = u >> k;
        const uint32_t rem = k ? (u & ((1u << k) - 1u)) : 0u;
        if (q + 1u + k <= 32u) {
            put((((1u << q) - 1u) << (k + 1u)) | rem, q + 1u + k);
            return;
        }
        while (q >= 32u) {
            put(0xFFFFFFFFu, 32u);
            q -= 32u;
        }
        put((1u << q) - 1u, q);
        put(rem, k + 1u);
    }
    LACX_HD void finish() {  // the last, partial word is shared with the next thread
        if (fill) flush_word((uint32_t)(acc >> 32), true);
    }
};

LACX_HD uint64_t rice_len(uint32_t u, uint32_t k) { return (uint64_t)(u >> k) + 1u + k; }

// Rice code: q ones, a zero, k remainder bits (ref rice.cpp:17-32, block/encoder.cpp:79-87); k <= 31
template <class Or>
LACX_HD uint64_t put_rice(const BitTile* tile, uint64_t pos, uint32_t u, uint32_t k, Or&& or_word) {
    const uint32_t q = u >> k;
    if (tile) {
        put_ones(tile, pos, q, or_word);
        const uint32_t rem = k ? (u & ((1u << k) - 1u)) : 0u;
        put_bits(tile, pos + q, rem, k + 1u, or_word);  // the zero stop bit rides on top of the remainder
    }
    return (uint64_t)q + 1u + k;
}

// One walk over the thread's chunk.  PASS 1 (tile == nullptr): computes kin per sample (stored in
// sh.xp.o.kin) and returns the thread's token bits.  PASS 2: writes the tokens starting at bit `pos`.
// The walk mirrors phase_b (stateful, p == 0) / partition_pass (stateless, p > 0).
template <class G, bool NARROW, bool PASS1, class M, class Or, class St>
LACX_HD uint64_t emit_walk_t(const Thread<G>& th, M& sh, const BitTile* tile_in, uint64_t pos, Or&& or_word,
                             St&& st_word) {
    if (th.cnt <= 0) return 0;
    const BitTile* tile = PASS1 ? nullptr : tile_in;
    TokenSink<Or, St> sink(tile ? *tile : BitTile{nullptr, 0, 0}, pos, or_word, st_word);
    const int t = th.tid;
    const uint32_t n = th.n;
    const uint32_t p = sh.p, parts = sh.parts;
    constexpr bool pass1 = PASS1;
    const uint32_t base = p ? (n >> p) : n;
    uint32_t part = p ? ((uint32_t)th.a / base) : 0u;
    if (part >= parts) part = parts - 1u;
    uint32_t s = part * base;
    uint32_t e = (part + 1u == parts) ? n : s + base;
    uint32_t mode = sh.part_mode_k[part] >> 5, k0 = sh.part_mode_k[part] & 31u;
    if (PASS1) {
        // Static-Rice partitions (the usual choice: the mode is preferred within 5 %, ref block/encoder.cpp:518) need no
        // walk: the parameter is the partition's for every sample and the chunk's token bits follow from the thread's
        // bit-sliced plane counts, sum (u >> k) = sum over the slices l of (cs[l] >> k) << l.  Taken when every lane of
        // the wave has a complete chunk inside one static partition.
        const bool whole_static = th.cnt == G::CH && (uint32_t)th.a + (uint32_t)G::CH <= e && mode == 3u;
        if (wave_all(whole_static)) {
            uint64_t shifted = 0;
#pragma unroll
            for (int l = 0; l < G::LV; ++l) shifted += (uint64_t)(th.cs[l] >> k0) << l;
#pragma unroll
            for (int i = 0; i < G::CH; ++i) sh.xp.o.kin[i * G::T + t] = (uint8_t)k0;
            return shifted + (uint64_t)G::CH * (1u + k0);
        }
    }
    // A chunk of nothing but zeros in the middle of a zero run -- the run began before the chunk inside the same
    // zero-run partition and the chunk does not reach the partition's end -- emits nothing: the run's one token went out
    // at the run's first sample.  Known from the block scans alone (no nonzero sample in the chunk: the "last nonzero
    // before" of the next chunk is this chunk's), so neither walk touches a sample.  Digital silence and sparse material
    // are made of such chunks; the general walk below spent 20 000 - 50 000 cycles per slot on them.
    if (mode == 1u && th.cnt == G::CH && (uint32_t)th.a > s && (uint32_t)th.a + (uint32_t)G::CH <= e) {
        const int32_t nz_before = sh.tabNZ[t], nz_through = sh.tabNZ[t + 1];
        if (nz_through == nz_before && nz_before < th.a - 1) return 0;
    }
    // --- state for the Rice parameter (pass 1 only) ---
    uint64_t P = sh.tabP[t];
    uint64_t Pseg = 0, W = 0;
    uint32_t D = 0, kin = k0;  // D: packed flag counts over the last 96 samples
    const bool stateful = (p == 0);
    const uint32_t m256 = (t >= G::W256) ? 0xFFFFFFFFu : 0u;
    const int t256 = (t >= G::W256) ? t - G::W256 : t, t96 = (t >= G::W96) ? t - G::W96 : t;
    // micro-window flags entering / leaving the 96-window (stateful walk only; layout of the packed counts, see phase_b_span)
    const uint32_t fin = (pass1 && p == 0u) ? ((uint32_t)sh.tabLM[t] | ((uint32_t)sh.tabZM[t] << 16)) : 0u;
    const uint32_t fout = (pass1 && p == 0u && t >= G::W96) ? ((uint32_t)sh.tabLM[t96] | ((uint32_t)sh.tabZM[t96] << 16)) : 0u;
    uint32_t c = (uint32_t)th.a;
    if (pass1) {
        if (stateful) {
            W = (t >= G::W256) ? sh.tabP[t - G::W256] : 0;
            D = window_flags<G>(sh.tabF, t);
            if (th.a > 0 && mode != 3u) kin = biased_k<NARROW>(kmean_t<NARROW>(P, c), P, W, D, c);
        } else {
            const uint32_t cs = s / G::CH;
            Pseg = sh.tabP[cs];
            for (uint32_t j = cs * G::CH; j < s; ++j) Pseg += sh.u[sw<G>((int)j)] & 0x3FFFFFFFu;
            if ((uint32_t)th.a > s && mode != 3u) kin = kmean_t<NARROW>(P - Pseg, (uint32_t)th.a - s);
        }
    }
    // --- zero-run structure ---
    int32_t lastnz = sh.tabNZ[t];
    if (lastnz < (int32_t)s - 1) lastnz = (int32_t)s - 1;
    int32_t f = th.a - 1 - lastnz;
    uint32_t w0 = sh.u[t];
    uint32_t x1 = peek_u<G>(sh, (uint32_t)th.a + 1u, n), x2 = peek_u<G>(sh, (uint32_t)th.a + 2u, n),
             x3 = peek_u<G>(sh, (uint32_t)th.a + 3u, n);
    uint64_t bits = 0;
    for (int i = 0; i < th.cnt; ++i) {
        const uint32_t j = (uint32_t)(th.a + i);
        if (j == e) {  // partition boundary inside the chunk
            ++part;
            s = e;
            e = (part + 1u == parts) ? n : s + base;
            mode = sh.part_mode_k[part] >> 5;
            k0 = sh.part_mode_k[part] & 31u;
            Pseg = P;
            f = 0;
            kin = k0;
        }
        const uint32_t u = w0 & 0x3FFFFFFFu;
        if (pass1) {
            if (mode == 3u || j == s) kin = k0;
            sh.xp.o.kin[i * G::T + t] = (uint8_t)kin;
        } else {
            kin = sh.xp.o.kin[i * G::T + t];
        }
        // ---- token ----
        uint64_t len = 0;
        if (mode == 0u || mode == 3u) {
            if (tile) sink.put_rice(u, kin);
            len = rice_len(u, kin);
        } else if (mode == 2u) {  // bin (ref block/encoder.cpp:609-667)
            if (u == 0u) {
                if (tile) sink.put(0u, 2u);
                len = 2;
            } else if (u <= 4u) {  // |v| = 1 -> tag 01, |v| = 2 -> tag 10, then the sign (u odd <=> negative)
                if (tile) sink.put(((u <= 2u ? 1u : 2u) << 1) | (u & 1u), 3u);
                len = 3;
            } else {
                if (tile) {
                    sink.put(3u, 2u);
                    sink.put_rice(u, kin);
                }
                len = 2u + rice_len(u, kin);
            }
        } else {  // zero-run (ref block/encoder.cpp:669-771)
            const bool z = (u == 0u);
            f = z ? f + 1 : 0;
            const uint32_t n1 = (j + 1u < e) ? x1 : 1u, n2 = (j + 2u < e) ? x2 : 1u, n3 = (j + 3u < e) ? x3 : 1u;
            const int ahead = (n1 != 0) ? 0 : ((n2 != 0) ? 1 : ((n3 != 0) ? 2 : 3));
            const bool in4 = z && (f + ahead >= 4);
            if (!in4) {
                const uint32_t esc = 1u << ((kin + 3u) < 24u ? (kin + 3u) : 24u);
                if (u > esc) {
                    if (tile) {
                        sink.put(2u, 2u);
                        sink.put(u, 32u);
                    }
                    len = 34;
                } else {
                    if (tile) {
                        sink.put(0u, 2u);
                        sink.put_rice(u, kin);
                    }
                    len = 2u + rice_len(u, kin);
                }
            } else if (f == 1) {  // first sample of a run of >= 4 zeros: one token for the whole run
                // run length: up to the next non-zero sample or the end of the partition
                uint32_t nx = n;
                for (int q = i + 1; q < th.cnt; ++q) {
                    if ((sh.u[q * G::T + t] & 0x3FFFFFFFu) != 0) {
                        nx = (uint32_t)(th.a + q);
                        break;
                    }
                }
                if (nx == n) nx = (uint32_t)sh.tabNX[t];
                const uint32_t run = (nx < e ? nx : e) - j;
                if (tile) {
                    sink.put(1u, 2u);
                    sink.put_rice(run - 4u, 2u);
                }
                len = 2u + rice_len(run - 4u, 2u);
            }
        }
        if (mode != 1u) {  // keep the run counter coherent when modes change between partitions
            f = (u == 0u) ? f + 1 : 0;
        }
        bits += len;
        // ---- state after this sample (pass 1) ----
        if (pass1) {
            P += u;
            ++c;
            if (stateful) {
                W += sh.u[i * G::T + t256] & m256;
                D += (fin >> i) & 0x00010001u;
                D -= (fout >> i) & 0x00010001u;
                kin = biased_k<NARROW>(kmean_t<NARROW>(P, c), P, W, D, c);
            } else {
                kin = kmean_t<NARROW>(P - Pseg, j + 1u - s);
            }
        }
        w0 = sh.u[((i + 1) & (G::CH - 1)) * G::T + t];
        x1 = x2;
        x2 = x3;
        x3 = peek_u<G>(sh, j + 4u, n);
    }
    if (tile) sink.finish();
    return bits;
}

// PASS 1 when tile == nullptr, PASS 2 otherwise (the second pass does not depend on the prefix-sum width)
template <class G, bool NARROW, class M, class Or, class St>
LACX_HD uint64_t emit_walk(const Thread<G>& th, M& sh, const BitTile* tile, uint64_t pos, Or&& or_word,
                           St&& st_word) {
    if (tile == nullptr) return emit_walk_t<G, NARROW, true>(th, sh, nullptr, pos, or_word, st_word);
    return emit_walk_t<G, true, false>(th, sh, tile, pos, or_word, st_word);
}

// Header fields (ref block/encoder.cpp:773-795), spread over the first threads.
template <class G, class M, class Or>
LACX_HD void emit_header(const Thread<G>& th, M& sh, const BitTile* tile, Or&& or_word) {
    const uint32_t t = (uint32_t)th.tid;
    const uint32_t lpc_bits = sh.ptype == 2 ? 16u * sh.order : 0u;
    if (t == 0) {
        put_bits(tile, 0, sh.ptype, 8u, or_word);
        put_bits(tile, 8, sh.order, 8u, or_word);
        uint32_t control = ((uint32_t)(sh.part_mode_k[0] >> 5) & 3u) << 5;
        if (sh.p) control |= 0x80u | (sh.p & 0x0Fu);
        put_bits(tile, 16u + lpc_bits, control, 8u, or_word);
    }
    if (sh.ptype == 2 && t < sh.order) {
        put_bits(tile, 16u + 16u * t, (uint32_t)(uint16_t)sh.lpc.coef[0][t + 1], 16u, or_word);
    }
    if (t < sh.parts) put_bits(tile, 24u + lpc_bits + 7u * t, sh.part_mode_k[t] & 0x7Fu, 7u, or_word);
}

}  // namespace lacx
