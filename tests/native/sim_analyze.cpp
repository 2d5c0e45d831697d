// tests/native/sim_analyze.cpp -- TEST INFRASTRUCTURE: lock-step host simulation of the analysis kernel.
//
// Runs the very per-thread phase functions the HIP kernel is built from (csrc/analyze_core.h), one
// "thread" after the other between the points where the kernel has a workgroup barrier, with the
// cross-thread steps (block scans, plane-count reductions, LDS atomics) replaced by their sequential
// definitions.  This lets the CPU test-suite check the data-parallel reformulation against the oracle
// without a GPU.  It is not part of the product and is not a fallback: nothing under
// lossless-audio-codec_amd/ links or loads it.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "analyze_core.h"
#include "emit_core.h"

using namespace lacx;

namespace {

void autocorr13(const int32_t* x, uint32_t n, int64_t* r) {
    for (int k = 0; k <= 12; ++k) {
        int64_t s = 0;
        for (uint32_t i = (uint32_t)k; i < n; ++i) s += (int64_t)x[i] * (int64_t)x[i - (uint32_t)k];
        r[k] = s;
    }
}

template <class G>
void scans_after_r(Smem<G>& sh) {
    uint64_t run = 0;
    int32_t mx = -1;
    for (int t = 0; t < G::T; ++t) {
        const uint64_t v = sh.tabP[t];
        const int32_t z = sh.tabNZ[t];
        sh.tabP[t] = run;
        sh.tabNZ[t] = mx;
        run += v;
        if (z > mx) mx = z;
    }
    sh.tabP[G::T] = run;
    sh.tabNZ[G::T] = mx;
}

template <class G>
void plane_totals(Smem<G>& sh, const std::vector<Thread<G>>& th) {
    for (int b = 0; b < 32; ++b) sh.planeTot[0][b] = sh.planeTot256[0][b] = 0;
    for (int t = 0; t < G::T; ++t) {
        for (int l = 0; l < G::LV; ++l) {
            for (int b = 0; b < 30; ++b) {
                const uint32_t bit = (th[t].cs[l] >> b) & 1u;
                sh.planeTot[0][b] += bit << l;
                if (t < G::W256) sh.planeTot256[0][b] += bit << l;
            }
        }
    }
}

template <class G>
int run_sim(const int32_t* x, uint32_t n, int zero_run, int partitioning, int force_wide, ChannelPlan* out) {
    if (n == 0 || n > (uint32_t)G::MAXN) return 1;
    Smem<G>* shp = new Smem<G>;
    Smem<G>& sh = *shp;
    std::memset(shp, 0, sizeof(Smem<G>));
    std::vector<Thread<G>> th(G::T);
    SlotSrc src{x, nullptr, CH_L};
    for (int t = 0; t < G::T; ++t) {
        thread_init(th[t], n, t);
        stage_samples(th[t], sh, src, 0);
    }
    int64_t r[13];
    autocorr13(x, n, r);
    const int max_valid_order = (n > 1) ? (int)((n - 1 < 32u) ? n - 1 : 32u) : 0;
    levinson_candidates(r, max_valid_order, sh.lpc.coef, sh.lpc.used);
    sh.best_cand = -1;
    // pass 1: the pruning bounds of all candidates from one walk per thread (pass1_bounds), as in the kernel
    uint64_t cand_key[11];
    {
        uint32_t g[11] = {0}, nz[11] = {0}, n4[11] = {0}, ends[11] = {0};
        for (int t = 0; t < G::T; ++t) {
            BoundPartials bp[11];
            if (n == (uint32_t)G::MAXN) pass1_bounds<G, true>(th[t], sh, false, bp); else pass1_bounds<G, false>(th[t], sh, false, bp);
            const uint32_t beyond = (uint32_t)(G::CH - th[t].cnt);
            for (int c = 0; c < 11; ++c) {
                g[c] += 34u * (uint32_t)G::CH - bp[c].msum - bp[c].nz - beyond;
                nz[c] += bp[c].nz - beyond;
                n4[c] += bp[c].n4;
                ends[c] += bp[c].ends;
            }
        }
        for (int c = 0; c < 11; ++c) {
            const bool avail = !(c >= 6 && sh.lpc.used[c - 6] == 0);
            cand_key[c] = avail ? ((candidate_lower_bound(g[c], nz[c] + (n4[c] << 16), ends[c], n, zero_run) << 4) | (uint64_t)c) : ~0ull;
        }
    }
    // pass 2: exact costs in ascending (bound, index) order; the first candidate that cannot win ends the search
    uint32_t tried = 0;
    for (;;) {
        uint64_t best_key = ~0ull;
        for (int c = 0; c < 11; ++c)
            if (!((tried >> c) & 1u) && cand_key[c] < best_key) best_key = cand_key[c];
        if (best_key == ~0ull) break;
        const int cand = (int)(best_key & 15u);
        if (!(force_wide & 4) && candidate_pruned(best_key >> 4, cand, sh.best_bits, sh.best_cand)) break;
        tried |= 1u << cand;
        for (int t = 0; t < G::T; ++t) phase_r(th[t], sh, cand);
        scans_after_r(sh);
        plane_totals(sh, th);
        const uint32_t k0 = initial_k_from_planes(sh.planeTot256[0], n);
        const bool narrow = !(force_wide & 1) && sh.tabP[G::T] < kNarrowLimit;
        for (int t = 0; t < G::T; ++t) {
            if (narrow) phase_a<G, true>(th[t], sh); else phase_a<G, false>(th[t], sh);
        }
        sh.acc[0][0] = sh.acc[0][1] = sh.acc[0][2] = sh.acc[0][3] = 0;
        bool zr = false;
        for (int t = 0; t < G::T; ++t) zr = zr || th[t].has4 != 0;
        zr = zr && zero_run;
        static const bool qstats = getenv("LACX_SIM_QSTATS") != nullptr;
        uint32_t nquick = 0, nchunks = 0;
        const bool full = n == (uint32_t)G::MAXN;
        // The kernel's order: the costs of a chunk without the walk where that is provably the same thing
        // (phase_b_quick); the other chunks of waves 1.. are queued and walked afterwards by whichever lane the queue
        // hands them to; wave 0's chunks are always walked in place.
        sh.bqcount = 0;
        for (int t = 0; t < G::T; ++t) {
            const bool live = (uint32_t)th[t].a < n;
            bool quick = false;
            if (G::T > 64 && t >= 64 && live && !(force_wide & 16)) quick = phase_b_quick_dispatch<G>(th[t], sh, narrow, zr);
            if (quick && (force_wide & 32)) {  // self-check: the walk must agree
                const unsigned long long qr = th[t].crice, qb = th[t].cbin, qz = th[t].czr;
                const uint32_t qh = th[t].chasrun;
                phase_b_dispatch<G>(th[t], sh, k0, narrow, zr, full);
                if (qr != th[t].crice || qb != th[t].cbin || (zr && (qz != th[t].czr || qh != th[t].chasrun))) {
                    fprintf(stderr, "phase_b_quick disagrees with the walk: cand %d chunk %d\n", cand, t);
                    delete shp;
                    return 2;
                }
            }
            if (!quick) {
                th[t].crice = th[t].cbin = th[t].czr = 0;
                th[t].chasrun = 0;
                if (live && G::T > 64 && t >= 64 && !(force_wide & 16)) sh.bqueue[sh.bqcount++] = (uint16_t)t;
                else if (live) phase_b_dispatch<G>(th[t], sh, k0, narrow, zr, full);
            }
            nchunks += live;
            nquick += quick;
        }
        for (uint32_t e = 0; e < sh.bqcount; ++e) {  // lane e % 64 of some wave walks queued chunk e
            Thread<G>& lane = th[64 + (e % (uint32_t)(G::T > 64 ? G::T - 64 : 1))];
            phase_b_queued<G>(lane, sh, (int)sh.bqueue[e], k0, narrow, zr, full);
        }
        for (int t = 0; t < G::T; ++t) {
            sh.acc[0][0] += th[t].crice;
            sh.acc[0][1] += th[t].cbin;
            sh.acc[0][2] += th[t].czr;
            sh.acc[0][3] += th[t].chasrun;
        }
        if (qstats) fprintf(stderr, "phase_b: n=%u cand %d quick chunks %u of %u, queued %u\n", n, cand, nquick, nchunks, sh.bqcount);
        score_candidate(sh, cand, n, zero_run, k0, sh.planeTot[0], sh.acc[0]);
    }
    // partition search on the winner
    const int best = sh.best_cand;
    for (int t = 0; t < G::T; ++t) phase_r(th[t], sh, best);
    scans_after_r(sh);
    const bool pnarrow = !(force_wide & 1) && sh.tabP[G::T] < kNarrowLimit;
    PartMem<G>& pm = sh.xp.part;
    int max_p = 0;
    if (partitioning && n >= (uint32_t)kMinPartition) max_p = max_partition_order(n);
    if (max_p > 0) {
        for (int w = 0; w < 15; ++w)
            for (int g = 0; g <= G::NG; ++g) pm.grp[w][g] = 0;
        for (int t = 0; t < G::T; ++t) {
            uint32_t words[15];
            packed_planes(th[t], words);
            for (int w = 0; w < 15; ++w) pm.grp[w][t / G::TPG] += words[w];
        }
        for (int w = 0; w < 15; ++w) {
            uint32_t run = 0;
            for (int g = 0; g <= G::NG; ++g) {
                const uint32_t v = pm.grp[w][g];
                pm.grp[w][g] = run;
                run += v;
            }
        }
        for (int p = 1; p <= max_p; ++p)
            for (uint32_t part = 0; part < (1u << p); ++part) seg_static_eval(sh, n, p, part);
        for (int i = 0; i < G::NSEG; ++i) {
            pm.segacc[i][0] = pm.segacc[i][1] = pm.segacc[i][2] = 0;
            pm.segrun[i] = 0;
        }
        auto flush = [&](uint32_t idx, unsigned long long rc, unsigned long long bn, unsigned long long zr,
                         uint32_t hr) {
            pm.segacc[idx][0] += rc;
            pm.segacc[idx][1] += bn;
            pm.segacc[idx][2] += zr;
            pm.segrun[idx] |= hr;
        };
        auto flushq = [&](int, uint32_t idx, unsigned long long rc, unsigned long long bn, unsigned long long zr, uint32_t hr) {
            flush(idx, rc, bn, zr, hr);
        };
        const bool fused = pnarrow && !(force_wide & 2) && partitions_chunk_aligned<G>(n, max_p);
        const bool quick = fused && !(force_wide & 8);
        const bool with_zr = zero_run && sh.best_hasrun;
        pm.qcount = 0;
        if (getenv("LACX_SIM_QSTATS")) fprintf(stderr, "partition: n=%u narrow %d hasrun %d fused %d quick %d total_u %llu\n", n, (int)pnarrow, (int)sh.best_hasrun, (int)fused, (int)quick, (unsigned long long)sh.tabP[G::T]);
        auto enqueue = [&pm](uint32_t entry, bool ambiguous) { if (ambiguous) pm.queue[pm.qcount++] = (uint16_t)entry; };
        for (int w0 = 0; quick && w0 < G::T; w0 += 64) {
            // the kernel's default: no sample walk where the Rice parameter is constant over the chunk -- wave by wave; with
            // zero-run costs a wave with too many ambiguous pairs walks all orders at once instead
            if (!with_zr) {
                for (int t = w0; t < w0 + 64 && t < G::T; ++t) partition_quick<G>(th[t], sh, max_p, flushq, enqueue);
                continue;
            }
            std::vector<QuickPrep<G>> qp(64);
            uint32_t pairs = 0;
            for (int t = w0; t < w0 + 64 && t < G::T; ++t) {
                partition_quick_prepare<G>(th[t], sh, max_p, qp[t - w0]);
                pairs += (uint32_t)__builtin_popcount(qp[t - w0].amb);
            }
            for (int t = w0; t < w0 + 64 && t < G::T; ++t) {
                if (pairs > kQuickMaxPairs && !(force_wide & 64)) partition_fused<G, true>(th[t], sh, max_p, flushq);
                else partition_quick_costs<G>(th[t], sh, max_p, qp[t - w0], flushq, enqueue);
            }
        }
        for (int t = 0; !quick && t < G::T; ++t) {
            if (fused) {
                if (with_zr) partition_fused<G, true>(th[t], sh, max_p, flushq);
                else partition_fused<G, false>(th[t], sh, max_p, flushq);
                continue;
            }
            for (int p = 1; p <= max_p; ++p) {
                if (pnarrow) partition_pass<G, true>(th[t], sh, p, flush); else partition_pass<G, false>(th[t], sh, p, flush);
            }
        }
        for (uint32_t e = 0; quick && e < pm.qcount; ++e) {
            if (with_zr) partition_slow_entry<G, true>(sh, n, pm.queue[e], flush);
            else partition_slow_entry<G, false>(sh, n, pm.queue[e], flush);
        }
        if (quick && getenv("LACX_SIM_QSTATS")) fprintf(stderr, "quick: n=%u queued %u of %u\n", n, pm.qcount, (n / (uint32_t)G::CH) * (uint32_t)max_p);
        for (int p = 1; p <= max_p; ++p) {
            pm.pbits[p] = 0;
            const uint32_t segbase = (2u << (p - 1)) - 2u;
            for (uint32_t part = 0; part < (1u << p); ++part) pm.pbits[p] += seg_choose(sh, segbase + part, zero_run);
        }
    }
    std::memset(out, 0, sizeof(*out));
    finalize_plan(sh, n, zero_run, max_p, out);
    delete shp;
    return 0;
}

// Lock-step simulation of the device emit kernel: plan + samples -> exact channel-block bytes.
template <class G>
int run_emit_sim(const int32_t* x, uint32_t n, const ChannelPlan* plan, uint8_t* out, uint32_t cap) {
    if (n == 0 || n > (uint32_t)G::MAXN) return -1;
    EmitMem<G>* shp = new EmitMem<G>;
    EmitMem<G>& sh = *shp;
    std::memset(shp, 0, sizeof(EmitMem<G>));
    std::vector<Thread<G>> th(G::T);
    SlotSrc src{x, nullptr, CH_L};
    for (int t = 0; t < G::T; ++t) {
        thread_init(th[t], n, t);
        stage_samples(th[t], sh, src, 0);
    }
    emit_load_plan(sh, *plan);
    for (int t = 0; t < G::T; ++t) phase_r(th[t], sh, (int)sh.cand);
    {
        uint64_t run = 0;
        int32_t mx = -1;
        for (int t = 0; t < G::T; ++t) {
            const uint64_t v = sh.tabP[t];
            const int32_t z = sh.tabNZ[t];
            sh.tabP[t] = run;
            sh.tabNZ[t] = mx;
            run += v;
            if (z > mx) mx = z;
        }
        sh.tabP[G::T] = run;
    }
    for (int t = 0; t < G::T; ++t) emit_first_nonzero(th[t], sh);
    {
        int32_t mn = (int32_t)n;
        for (int t = G::T - 1; t >= 0; --t) {
            const int32_t v = sh.tabNX[t];
            sh.tabNX[t] = mn;
            if (v < mn) mn = v;
        }
    }
    const bool narrow = sh.tabP[G::T] < kNarrowLimit;
    if (sh.p == 0 && (sh.part_mode_k[0] >> 5) != 3) {
        for (int t = 0; t < G::T; ++t) {
            if (narrow) phase_a<G, true>(th[t], sh); else phase_a<G, false>(th[t], sh);
        }
    }
    // ownership check of the token writer: a plainly stored word must be untouched before and never OR-ed after
    std::vector<uint8_t> owned(kEmitTileWords, 0);
    int ownership_errors = 0;
    uint32_t* tile_base = sh.xp.o.obits;
    auto orw = [&](uint32_t* w, uint32_t v) {
        if (owned[w - tile_base]) ++ownership_errors;
        *w |= v;
    };
    auto stw = [&](uint32_t* w, uint32_t v) {
        if (*w != 0 || owned[w - tile_base]) ++ownership_errors;
        owned[w - tile_base] = 1;
        *w = v;
    };
    std::vector<uint64_t> bits(G::T), off(G::T);
    for (int t = 0; t < G::T; ++t)
        bits[t] = narrow ? emit_walk<G, true>(th[t], sh, nullptr, 0, orw, stw) : emit_walk<G, false>(th[t], sh, nullptr, 0, orw, stw);
    uint64_t total = sh.header_bits;
    for (int t = 0; t < G::T; ++t) {
        off[t] = total;
        total += bits[t];
    }
    const uint64_t bytes = (total + 7) / 8;
    int rc = (int)bytes;
    if (bytes != plan->payload_bytes) rc = -2;
    if (bytes > cap) rc = -3;
    if (rc >= 0) {
        for (uint64_t bit0 = 0; bit0 < bytes * 8; bit0 += (uint64_t)kEmitTileWords * 32) {
            for (int i = 0; i < kEmitTileWords; ++i) sh.xp.o.obits[i] = 0;
            std::fill(owned.begin(), owned.end(), 0);
            BitTile tile{sh.xp.o.obits, bit0, (uint32_t)kEmitTileWords};
            for (int t = 0; t < G::T; ++t) {
                emit_header(th[t], sh, &tile, orw);
                if (narrow) emit_walk<G, true>(th[t], sh, &tile, off[t], orw, stw); else emit_walk<G, false>(th[t], sh, &tile, off[t], orw, stw);
            }
            const uint64_t byte0 = bit0 / 8;
            for (uint64_t b = byte0; b < bytes && b < byte0 + (uint64_t)kEmitTileWords * 4; ++b) {
                const uint64_t rel = b - byte0;
                out[b] = (uint8_t)(sh.xp.o.obits[rel >> 2] >> (24 - 8 * (rel & 3)));
            }
        }
    }
    if (ownership_errors) rc = -4;
    delete shp;
    return rc;
}

}  // namespace

extern "C" {

// analysis + device-emit simulation of one channel block; returns bytes written or < 0
int sim_block_encode(const int32_t* x, uint32_t n, int zero_run, int partitioning, int force_wide, uint8_t* out,
                     uint32_t cap) {
    ChannelPlan plan;
    if (run_sim<Geo<16, 1024>>(x, n, zero_run, partitioning, force_wide, &plan) != 0) return -1;
    return run_emit_sim<Geo<16, 1024>>(x, n, &plan, out, cap);
}

// geo: 0 = <16,1024> (full blocks), 1 = <4,64> (probe windows)
// force_wide bit 0: run the 64-bit arithmetic variants even where the 32-bit fast path would be taken;
// bit 1: use the per-order partition passes even where the fused pass applies; bit 2: no candidate pruning;
// bit 3: the fused walk instead of partition_quick; bit 4: no phase_b_quick; bit 5: check every phase_b_quick result
// against the walk; bit 6: partition_quick for every wave, whatever its count of ambiguous pairs
int sim_block_plan(const int32_t* x, uint32_t n, int zero_run, int partitioning, int geo, int force_wide,
                   ChannelPlan* out) {
    if (geo == 0) return run_sim<Geo<16, 1024>>(x, n, zero_run, partitioning, force_wide, out);
    return run_sim<Geo<4, 64>>(x, n, zero_run, partitioning, force_wide, out);
}

// kmean() against the division it replaces; returns the number of mismatches.
uint64_t sim_kmean_check(uint64_t seed, uint64_t iters) {
    uint64_t bad = 0, s = seed * 0x9E3779B97F4A7C15ull + 1;
    for (uint64_t i = 0; i < iters; ++i) {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        const uint32_t c = 1u + (uint32_t)((s >> 8) % 16384u);
        const int sh = (int)((s >> 40) % 46u);
        uint64_t S = (s * 0xD6E8FEB86659FD93ull) >> (63 - sh);
        if (i % 7 == 0) S = (uint64_t)c * ((s >> 20) % 70000u) + ((s >> 3) % 3u) - 1u + (c >> 1);  // near k boundaries
        if ((int64_t)S < 0) S = 0;
        const uint64_t mean = (S + (c >> 1)) / c;
        uint32_t k = 0;
        if (mean > 1) {
            uint64_t v = mean - 1;
            while (v) {
                ++k;
                v >>= 1;
            }
            if (k > 31) k = 31;
        }
        if (mean < (1ull << 31) && kmean(S, c) != k) ++bad;
    }
    return bad;
}
}
