// api_core.cpp -- the encoder object behind the C ABI (include/lacx.h): device and stream set-up, the device workspace and
// the pinned buffers, argument validation, pipeline chunking, timing and small helpers shared by the other host units.
// There is no CPU analysis path here: without a usable HIP device every analysing call fails.
#include "encoder_impl.h"

namespace lacx_host {

Knobs read_knobs() {
    Knobs k;
    auto set = [](const char* name) {
        const char* v = std::getenv(name);
        return v != nullptr;
    };
    auto on = [](const char* name) {  // set, non-empty and not "0"
        const char* v = std::getenv(name);
        return v && *v && *v != '0';
    };
    auto num = [](const char* name) -> unsigned long long {
        const char* v = std::getenv(name);
        return v ? std::strtoull(v, nullptr, 0) : 0ull;
    };
    k.stream_priority = !set("LACX_NO_STREAM_PRIORITY");
    if (const char* v = std::getenv("LACX_FUSED_EMIT")) k.fused_emit = *v != '0';
    k.emit_staged = on("LACX_EMIT_STAGED");
    k.direct_packer = set("LACX_DIRECT_PACKER");
    k.packer = !set("LACX_NO_PACKER");
    k.chain = !set("LACX_NO_CHAIN");
    k.persistent = !set("LACX_NO_PERSISTENT");
    k.debug_drain = set("LACX_DEBUG_DRAIN");
    k.two_copy_streams = !set("LACX_ONE_COPY_STREAM");
    if (const char* v = std::getenv("LACX_STREAM_PRIO")) {  // "main,chunks,pack", each -1 (high) / 0 (normal) / 1 (low)
        int a = 0, b = 0, c = 0;
        if (std::sscanf(v, "%d,%d,%d", &a, &b, &c) == 3) {
            k.prio_main = a;
            k.prio_chunks = b;
            k.prio_pack = c;
        }
    }
    k.front_halves = !set("LACX_NO_FRONT_HALVES");
    k.tune.fold_front = !set("LACX_NO_FRONT_FOLD");
    k.lazy_repair = !set("LACX_NO_LAZY_REPAIR");
    k.silent_template = !set("LACX_NO_SILENT_TEMPLATE");
    k.front_stream_split = set("LACX_FRONT_STREAM");  // (measured: slower, see DESIGN 8 -- kept as an experiment switch)
    k.pinned_cap_bytes = num("LACX_PINNED_CAP_BYTES");
    k.debug_skip = (uint32_t)num("LACX_DEBUG_SKIP");
    k.pipe_chunks = (uint32_t)num("LACX_PIPE_CHUNKS");
    if (const char* v = std::getenv("LACX_PIPE_SPLIT")) k.pipe_split = v;
    k.drain_fence = (uint32_t)num("LACX_DRAIN_FENCE");
    if (const char* v = std::getenv("LACX_FANOUT_EXCHANGE")) k.fanout_exchange = std::strcmp(v, "host") == 0 ? 1u : (std::strcmp(v, "rccl") == 0 ? 2u : 0u);
    k.tune.persistent_grid = (uint32_t)num("LACX_PERSISTENT_GRID");
    k.tune.pack_nap = (int)num("LACX_PACK_NAP");
    k.tune.pack_grid = (int)num("LACX_PACK_GRID");
    k.tune.no_pairs = set("LACX_NO_PAIRS");
    return k;
}

double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

int fail(lacx_encoder* e, int code, const std::string& msg) {
    if (e) e->err = msg;
    return code;
}

int hip_fail(lacx_encoder* e, hipError_t err, const char* what) {
    return fail(e, LACX_E_DEVICE, std::string(what) + ": " + hipGetErrorString(err));
}

// Copy of a large result out of the pinned buffer: split over a few threads (a single memcpy into freshly allocated
// memory runs at page-fault speed; the tens of MB of a payload took longer than the whole device encode).
void big_copy(uint8_t* dst, const uint8_t* src, uint64_t n) {
    constexpr uint64_t kPiece = 4ull << 20;
    const unsigned hw = std::thread::hardware_concurrency();
    const uint64_t want = std::min<uint64_t>(std::min<uint64_t>(8, hw ? hw : 1), n / kPiece);
    if (want < 2) {
        std::memcpy(dst, src, n);
        return;
    }
    std::vector<std::thread> pool;
    const uint64_t per = (n / want + 4095) & ~4095ull;
    for (uint64_t t = 1; t < want; ++t) {
        const uint64_t o = t * per;
        if (o >= n) break;
        pool.emplace_back([=] { std::memcpy(dst + o, src + o, std::min(per, n - o)); });
    }
    std::memcpy(dst, src, std::min(per, n));
    for (auto& th : pool) th.join();
}

int ensure_device(lacx_encoder* e) {
    if (e->device_ready) return LACX_OK;
    int count = 0;
    const hipError_t ce = hipGetDeviceCount(&count);
    if (ce != hipSuccess || count <= 0)
        return fail(e, LACX_E_DEVICE, "no HIP device available: the LAC analysis path has no CPU fallback");
    int dev = e->cfg.device;
    if (dev < 0) HIP_TRY(e, hipGetDevice(&dev), "hipGetDevice");
    if (dev >= count) return fail(e, LACX_E_DEVICE, "HIP device ordinal out of range");
    HIP_TRY(e, hipSetDevice(dev), "hipSetDevice");
    e->device = dev;
    {
        // Stream priorities.  The HIP runtime multiplexes streams onto a small pool of hardware queues PER PRIORITY LEVEL,
        // and commands of streams that share a hardware queue execute in order.  The streaming packer must therefore be
        // alone on its level: behind the analysis kernel in a shared queue it starts when the analysis is over (every
        // encoder of a process but the first few: drain_first_ms 4.4 instead of 0.9 ms, steps 50 % longer -- the round-4
        // bench caught it); sharing a queue with the uploader's stream it deadlocks against the upload it waits for until
        // its 20 ms bound expires (a test caught that).  So: the analysis streams (pipeline chunk c on stream c) at HIGH
        // priority (front kernels measured 20 % slower at normal), the packer at LOW, where no other stream of this
        // library or, by default, of anybody else lives; copy, upload and decoder streams at the default (normal).
        // Measured (WAV -> .lac, 10 min stream, 4 chunks): high/high/low 3.42 ms, high/low/normal 3.40, high/normal/low
        // 3.73, all high (round 3) 3.41.  LACX_STREAM_PRIO=main,chunks,pack overrides.
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const bool prio = e->knobs.stream_priority && greatest < least;
        const int normal = (greatest + least) / 2 > greatest ? (greatest + least) / 2 : std::min(greatest + 1, least);
        auto level = [&](int code) { return code < 0 ? greatest : (code == 0 ? normal : least); };  // -1 high, 0 normal, 1 low
        int i = 0;
        for (auto& s : e->stream) {
            if (prio) {
                // (the last stream doubles as the lower-priority stream of the second half's front kernels,
                // LaunchTuning::aux_stream: NORMAL, not low -- the packer must stay alone on its level, see above; two
                // encoders on one device whose fourth pipeline chunk shared a hardware queue with a packer stalled 20 ms)
                const int p = level(i == 0 ? e->knobs.prio_main : (i == kStreams - 1 ? 0 : e->knobs.prio_chunks));
                HIP_TRY(e, hipStreamCreateWithPriority(&s, hipStreamNonBlocking, p), "hipStreamCreate");
            } else {
                HIP_TRY(e, hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
            }
            ++i;
        }
    }
    for (auto& row : e->ev)
        for (auto& ev : row) HIP_TRY(e, hipEventCreate(&ev), "hipEventCreate");
    for (auto& ev : e->done) HIP_TRY(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
    for (auto& ev : e->copied) HIP_TRY(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
    HIP_TRY(e, hipEventCreateWithFlags(&e->prologue, hipEventDisableTiming), "hipEventCreate");
    for (auto& ev : e->aux_ev) HIP_TRY(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
    HIP_TRY(e, hipHostMalloc((void**)&e->h_totals, sizeof(unsigned long long) * kMaxChunks, 0), "hipHostMalloc");
    HIP_TRY(e, hipHostMalloc((void**)&e->h_err, sizeof(uint32_t) * (kMaxChunks + 8), 0), "hipHostMalloc");
    {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const int normal = (greatest + least) / 2 > greatest ? (greatest + least) / 2 : std::min(greatest + 1, least);
        const int pp = e->knobs.prio_pack < 0 ? greatest : (e->knobs.prio_pack == 0 ? normal : least);
        HIP_TRY(e, hipStreamCreateWithPriority(&e->pack_stream, hipStreamNonBlocking, pp), "hipStreamCreate");
    }
    HIP_TRY(e, hipEventCreateWithFlags(&e->pack_done, hipEventDisableTiming), "hipEventCreate");
    HIP_TRY(e, hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking), "hipStreamCreate");
    HIP_TRY(e, hipStreamCreateWithFlags(&e->copy_stream2, hipStreamNonBlocking), "hipStreamCreate");
    HIP_TRY(e, hipHostMalloc((void**)&e->h_tspan, sizeof(unsigned long long) * 2 * kMaxChunks, 0), "hipHostMalloc");
    e->device_ready = true;
    return LACX_OK;
}

EmitPool& pool_of(lacx_encoder* e) {
    if (!e->pool) {
        unsigned nt = e->cfg.emit_threads ? e->cfg.emit_threads : std::thread::hardware_concurrency();
        if (nt == 0) nt = 1;
        e->pool.reset(new EmitPool(nt > 1 ? nt - 1 : 0));  // the calling thread is the last worker
    }
    return *e->pool;
}

void free_workspace(lacx_encoder* e) {
    if (e->ws.plans) (void)hipFree(e->ws.plans);
    if (e->ws.bplans) (void)hipFree(e->ws.bplans);
    if (e->ws.need_probe) (void)hipFree(e->ws.need_probe);
    if (e->ws.need_full) (void)hipFree(e->ws.need_full);
    if (e->ws.acorr) (void)hipFree(e->ws.acorr);
    if (e->ws.lpcs) (void)hipFree(e->ws.lpcs);
    if (e->ws.sums) (void)hipFree(e->ws.sums);
    if (e->ws.badidx) (void)hipFree(e->ws.badidx);
    if (e->ws.front_ctr) (void)hipFree(e->ws.front_ctr);
    if (e->ws.block_off) (void)hipFree(e->ws.block_off);
    if (e->ws.table) (void)hipFree(e->ws.table);
    if (e->ws.stream_pre) (void)hipFree(e->ws.stream_pre);
    if (e->zero_region) (void)hipFree(e->zero_region);  // size_rec, ready_rec, tspan, emitted, packed, err_flag
    e->zero_region = nullptr;
    e->d_tspan = nullptr;
    e->d_work_ctr = nullptr;
    e->d_silent = nullptr;
    e->ws = DeviceWorkspace{};
    e->ws_blocks = 0;
}

int ensure_workspace(lacx_encoder* e, uint32_t nblocks) {
    if (nblocks > e->ws_blocks) {
        free_workspace(e);
        const size_t slots = (size_t)nblocks * kSlotsPerBlock;
        HIP_TRY(e, hipMalloc((void**)&e->ws.plans, slots * sizeof(ChannelPlan)), "hipMalloc(plans)");
        HIP_TRY(e, hipMalloc((void**)&e->ws.bplans, (size_t)nblocks * sizeof(BlockPlan)), "hipMalloc(bplans)");
        HIP_TRY(e, hipMalloc((void**)&e->ws.need_probe, (size_t)nblocks * 4), "hipMalloc(need)");
        HIP_TRY(e, hipMalloc((void**)&e->ws.need_full, (size_t)nblocks * 4), "hipMalloc(need)");
        HIP_TRY(e, hipMalloc((void**)&e->ws.acorr, slots * 13 * sizeof(int64_t)), "hipMalloc(acorr)");
        HIP_TRY(e, hipMalloc((void**)&e->ws.lpcs, slots * sizeof(LpcSet)), "hipMalloc(lpcs)");
        HIP_TRY(e, hipMalloc((void**)&e->ws.sums, (size_t)nblocks * 12 * sizeof(unsigned long long)), "hipMalloc(sums)");
        HIP_TRY(e, hipMalloc((void**)&e->ws.badidx, (size_t)nblocks * 2 * sizeof(uint32_t)), "hipMalloc(badidx)");
        // (zeroed once: whoever completes a count puts the word back to zero)
        HIP_TRY(e, hipMalloc((void**)&e->ws.front_ctr, (size_t)nblocks * 2 * sizeof(uint32_t)), "hipMalloc(front counters)");
        HIP_TRY(e, hipMemset(e->ws.front_ctr, 0, (size_t)nblocks * 2 * sizeof(uint32_t)), "hipMemset(front counters)");
        // (the null stream's memset is not ordered against this encoder's non-blocking streams: wait for it here, once)
        HIP_TRY(e, hipStreamSynchronize(nullptr), "hipStreamSynchronize");
        HIP_TRY(e, hipMalloc((void**)&e->ws.block_off, ((size_t)nblocks + kMaxChunks + 1) * sizeof(unsigned long long)),
                "hipMalloc(block_off)");
        HIP_TRY(e, hipMalloc((void**)&e->ws.table, (size_t)nblocks * 2 * sizeof(uint32_t)), "hipMalloc(table)");
        HIP_TRY(e, hipMalloc((void**)&e->ws.stream_pre, ((size_t)nblocks + 1) * sizeof(unsigned long long)), "hipMalloc(stream prefixes)");

        // Everything a call needs zeroed up front lives in ONE allocation, cleared by one memset: the hand-off records
        // and flags of the fused emit (per channel block of the shard), the error flags, the kernel time stamps.
        {
            const size_t items = (size_t)nblocks * 2 + 4;
            const size_t ranges = items / kPackerRangeItems + 2;  // packer progress (copy-engine drain)
            const size_t bytes = items * (2 * sizeof(unsigned long long) + 2 * sizeof(uint32_t)) +
                                 sizeof(unsigned long long) * 2 * kMaxChunks + sizeof(uint32_t) * (kMaxChunks + 8) +
                                 ranges * (sizeof(unsigned long long) + sizeof(uint32_t)) + 16 +
                                 sizeof(uint32_t) * 8 * (kMaxChunks + 1) +  // work counters of the persistent analysis
                                 16 + sizeof(SilentTemplate);
            e->zero_bytes = (bytes + 15) & ~(size_t)15;
            HIP_TRY(e, hipMalloc((void**)&e->zero_region, e->zero_bytes), "hipMalloc(zeroed region)");
            uint8_t* p = e->zero_region;
            e->ws.size_rec = reinterpret_cast<unsigned long long*>(p);
            p += items * sizeof(unsigned long long);
            e->ws.ready_rec = reinterpret_cast<unsigned long long*>(p);
            p += items * sizeof(unsigned long long);
            e->d_tspan = reinterpret_cast<unsigned long long*>(p);
            p += sizeof(unsigned long long) * 2 * kMaxChunks;
            e->ws.emitted = reinterpret_cast<uint32_t*>(p);
            p += items * sizeof(uint32_t);
            e->ws.packed = reinterpret_cast<uint32_t*>(p);
            p += items * sizeof(uint32_t);
            e->ws.err_flag = reinterpret_cast<uint32_t*>(p);
            p += sizeof(uint32_t) * (kMaxChunks + 8);
            p = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(p) + 7) & ~(uintptr_t)7);
            e->d_range_end = reinterpret_cast<unsigned long long*>(p);
            p += ranges * sizeof(unsigned long long);
            e->d_range_cnt = reinterpret_cast<uint32_t*>(p);
            p += ranges * sizeof(uint32_t);
            p = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(p) + 3) & ~(uintptr_t)3);
            e->d_work_ctr = reinterpret_cast<uint32_t*>(p);
            p += sizeof(uint32_t) * 8 * (kMaxChunks + 1);
            p = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(p) + 15) & ~(uintptr_t)15);
            e->d_silent = reinterpret_cast<SilentTemplate*>(p);
        }
        e->ws_blocks = nblocks;
    }
    if (nblocks > e->h_blocks) {
        if (e->h_plans) (void)hipHostFree(e->h_plans);
        if (e->h_bplans) (void)hipHostFree(e->h_bplans);
        e->h_plans = nullptr;
        e->h_bplans = nullptr;
        e->h_blocks = 0;
        HIP_TRY(e, hipHostMalloc((void**)&e->h_plans, (size_t)nblocks * kSlotsPerBlock * sizeof(ChannelPlan), 0),
                "hipHostMalloc(plans)");
        HIP_TRY(e, hipHostMalloc((void**)&e->h_bplans, (size_t)nblocks * sizeof(BlockPlan), 0),
                "hipHostMalloc(bplans)");
        e->h_blocks = nblocks;
    }
    return LACX_OK;
}

// Staging slots of the fused emit: one per channel block, fixed stride.  3 bytes per sample cover any 16-bit material
// and 5 any 24-bit material with room to spare (raw PCM is 2 resp. 3; the costliest constructible streams stay under
// 1.2 x raw); a longer bitstream simply falls back to k_emit.  Device memory only: 48 KiB per channel block of 16-bit
// audio (a 2 h stereo shard: 2 GB of the 288 GB).
int ensure_slots(lacx_encoder* e, uint32_t nblocks, int channels, int bit_depth) {
    const unsigned long long stride = (unsigned long long)kMaxBlock * ((bit_depth ? bit_depth : e->cfg.bit_depth) == 16 ? 3u : 5u);
    const unsigned long long need = stride * nblocks * (unsigned)channels + 64u;
    if (need > e->slots_cap) {
        if (e->slots) (void)hipFree(e->slots);
        e->slots = nullptr;
        e->slots_cap = 0;
        HIP_TRY(e, hipMalloc((void**)&e->slots, need), "hipMalloc(emit slots)");
        e->slots_cap = need;
    }
    e->ws.slots = e->slots;
    e->ws.slot_stride = stride;
    return LACX_OK;
}

int ensure_pcm(lacx_encoder* e, uint64_t frames, bool stereo) {
    if (frames > e->d_cap || (stereo && !e->d_right)) {
        if (e->d_left) (void)hipFree(e->d_left);
        if (e->d_right) (void)hipFree(e->d_right);
        e->d_left = e->d_right = nullptr;
        e->d_cap = 0;
        HIP_TRY(e, hipMalloc((void**)&e->d_left, frames * sizeof(int32_t)), "hipMalloc(left)");
        HIP_TRY(e, hipMalloc((void**)&e->d_right, frames * sizeof(int32_t)), "hipMalloc(right)");
        e->d_cap = frames;
    }
    return LACX_OK;
}

bool rate_ok(uint32_t sr) { return sr == 44100 || sr == 48000 || sr == 96000 || sr == 192000; }

// Argument validation of LAC::Encoder::encode (ref lac/encoder.cpp:220-237), same order and wording.
int validate_stream_args(lacx_encoder* e, const void* left, uint64_t frames) {
    if (left == nullptr || frames == 0) return fail(e, LACX_E_INVALID, "left channel must not be empty");
    if (!rate_ok(e->cfg.sample_rate))
        return fail(e, LACX_E_INVALID, "unsupported sample rate: " + std::to_string(e->cfg.sample_rate));
    if (!(e->cfg.bit_depth == 16 || e->cfg.bit_depth == 24))
        return fail(e, LACX_E_INVALID, "unsupported bit depth: " + std::to_string((int)e->cfg.bit_depth));
    if (e->cfg.stereo_mode > 2)
        return fail(e, LACX_E_INVALID, "unsupported stereo mode: " + std::to_string((int)e->cfg.stereo_mode));
    return LACX_OK;
}

uint32_t blocks_for(uint64_t frames) { return (uint32_t)((frames + kMaxBlock - 1) / kMaxBlock); }

AnalyzeParams make_params(const lacx_encoder* e, uint64_t frames, int channels, int stereo_mode, int bit_depth,
                          int layout) {
    AnalyzeParams prm{};
    prm.layout = layout;
    prm.frames = frames;
    prm.num_blocks = blocks_for(frames);
    prm.first_block = 0;
    prm.channels = channels;
    prm.stereo_mode = channels == 2 ? stereo_mode : 0;
    prm.bit_depth = bit_depth;
    prm.zero_run = e->cfg.zero_run_enabled ? 1 : 0;
    prm.partitioning = e->cfg.partitioning_enabled ? 1 : 0;
    prm.debug_skip = e->knobs.debug_skip;  // test hooks / ablations (only a -DLACX_TEST_HOOKS library looks at it)
    return prm;
}

// Launch set of one stream (or one pipeline chunk of it): the descriptor travels in the kernel arguments.
LaunchSet one_stream_set(const AnalyzeParams& prm, const int32_t* left, const int32_t* right, uint32_t fuse_items,
                         uint64_t out_cap) {
    StreamDesc sd{};
    sd.prm = prm;
    sd.left = left;
    sd.right = right;
    sd.first_block = 0;
    sd.first_wg = 0;
    sd.fuse_items = fuse_items;
    sd.out_base = 0;
    sd.out_cap = out_cap;
    return single_set(sd);
}
// (LaunchSet::streams points into the set itself for one stream: fixed up wherever a set is copied or returned)
const LaunchSet& bind(LaunchSet& ls) {
    if (ls.br.table == nullptr) ls.streams = &ls.br.single;
    return ls;
}

DeviceWorkspace ws_at(const DeviceWorkspace& ws, uint32_t first_block) {
    DeviceWorkspace w = ws;
    const size_t s = (size_t)first_block * kSlotsPerBlock;
    w.plans += s;
    w.bplans += first_block;
    w.need_probe += first_block;
    w.need_full += first_block;
    w.acorr += s * 13;
    w.lpcs += s;
    w.sums += (size_t)first_block * 12;
    w.badidx += (size_t)first_block * 2;
    if (w.front_ctr) w.front_ctr += (size_t)first_block * 2;
    w.table += (size_t)first_block * 2;
    return w;
}

// Host emit wants many chunks (emit of chunk i overlaps the analysis of chunk i+1); with the emit on the
// device the only host work is a copy, and two chunks (payload copy of one under the kernels of the other)
// measured best.
std::vector<Chunk> plan_chunks(const Knobs& kn, uint32_t nb, bool device_emit, bool fused, bool upload) {
    uint32_t nchunks = nb / kMinChunkBlocks;
    // device emit without the fused path: 3 chunks up to an hour of stereo 48 kHz per call, 4 and 6 beyond (measured on a
    // 2 h shard).  With the fused emit + streaming packer nothing is left to overlap by chunking -- the payload leaves
    // while the analysis runs, and ingest / probes keep every CU busy by themselves -- and one launch set measured best
    // from 10 min to 2 h of audio (a chunked run only adds kernel boundaries).
    // With the input still in host memory the chunks pipeline the upload (the uploader thread copies chunk c + 1 while
    // chunk c's kernels are enqueued and run): four equal chunks measured best once the copies came from their own thread
    // and the packer's stream had a priority level of its own (10 min stream, WAV image -> .lac: 3.40 ms; 1:2:3 3.57,
    // 1:3:4 3.6, one chunk 4.65; round 3, copies issued by the calling thread: 1:3:4 3.75).
    const uint32_t dev_chunks = fused ? (upload ? 4u : 1u) : (nb >= 12000u ? 6u : (nb >= 6000u ? 4u : 3u));
    nchunks = std::max(1u, std::min(nchunks, device_emit ? dev_chunks : 8u));
    bool forced = false;
    if (kn.pipe_chunks >= 1 && kn.pipe_chunks <= (uint32_t)kMaxChunks) {  // tuning knob
        nchunks = std::min<uint32_t>(kn.pipe_chunks, nb);
        forced = true;
    }
    std::vector<Chunk> out;
    const char* split_env = kn.pipe_split.empty() ? nullptr : kn.pipe_split.c_str();  // tuning knob: relative chunk sizes, e.g. "5,3,1"
    // Device emit: three chunks on three streams of falling priority, the last one a little smaller -- its
    // emit is the only one whose PCIe writes are not hidden under another chunk's analysis (measured best).
    if (!split_env && !forced && device_emit && nchunks == 3u) split_env = (fused && upload) ? "1,2,3" : "5,5,4";
    if (const char* env = split_env) {
        std::vector<double> w;
        double sum = 0;
        for (const char* p = env; *p && w.size() < (size_t)kMaxChunks;) {
            char* end = nullptr;
            const double v = std::strtod(p, &end);
            if (end == p) break;
            if (v > 0) {
                w.push_back(v);
                sum += v;
            }
            p = (*end == ',') ? end + 1 : end;
        }
        if (!w.empty() && nb >= w.size()) {
            uint32_t f = 0;
            double acc = 0;
            for (size_t i = 0; i < w.size(); ++i) {
                acc += w[i];
                uint32_t end = i + 1 == w.size() ? nb : (uint32_t)(nb * (acc / sum));
                end = std::max(end, f + 1);
                end = std::min(end, nb - (uint32_t)(w.size() - 1 - i));
                out.push_back({f, end - f});
                f = end;
            }
            return out;
        }
    }
    const uint32_t per = (nb + nchunks - 1) / nchunks;
    for (uint32_t f = 0; f < nb; f += per) out.push_back({f, std::min(per, nb - f)});
    return out;
}

void add_chunk_timing(lacx_encoder* e, int c) {
    float f = 0;
    if (hipEventElapsedTime(&f, e->ev[c][0], e->ev[c][4]) == hipSuccess) e->timing.analysis_ms += f;
    if (hipEventElapsedTime(&f, e->ev[c][0], e->ev[c][1]) == hipSuccess) e->timing.ingest_ms += f;
    if (hipEventElapsedTime(&f, e->ev[c][1], e->ev[c][2]) == hipSuccess) e->timing.probe_ms += f;
    if (hipEventElapsedTime(&f, e->ev[c][2], e->ev[c][3]) == hipSuccess) e->timing.full_ms += f;
    (void)hipGetLastError();  // an event that was not recorded in this call must not poison the next launch check
}

void count_slots(lacx_encoder* e, uint32_t first, uint32_t count) {
    uint64_t fs = 0, ps = 0;
    for (uint32_t b = first; b < first + count; ++b) {
        const ChannelPlan* s = e->h_plans + (size_t)b * kSlotsPerBlock;
        for (int i = 0; i < 4; ++i) fs += s[i].valid;
        for (int i = 4; i < kSlotsPerBlock; ++i) ps += s[i].valid;
    }
    e->timing.full_slots += fs;
    e->timing.probe_slots += ps;
}

// Sample-range errors in the reference's order: all of left first, then right (ref lac/encoder.cpp:238-241).
int check_sample_range(lacx_encoder* e, uint32_t nb) {
    e->bad_channel = -1;
    for (int pass = 0; pass < 2; ++pass) {
        for (uint32_t b = 0; b < nb; ++b) {
            const BlockPlan& bp = e->h_bplans[b];
            if (!bp.invalid) continue;
            const bool is_right = (bp.first_bad >> 31) != 0;
            // per block the left channel wins the minimum, so a "right" entry means a clean left channel
            if ((pass == 0) == is_right) continue;
            const uint64_t idx = (uint64_t)b * kMaxBlock + (bp.first_bad & 0x7FFFFFFFu);
            e->bad_channel = is_right ? 1 : 0;
            e->bad_index = idx;
            return fail(e, LACX_E_INVALID,
                        std::string(is_right ? "right" : "left") + " sample at index " + std::to_string(idx) +
                            " is outside the configured PCM bit depth");
        }
    }
    return LACX_OK;
}

void reset_device_timing(lacx_encoder* e) {
    e->timing.analysis_ms = e->timing.ingest_ms = e->timing.probe_ms = e->timing.full_ms = 0;
    e->timing.full_slots = e->timing.probe_slots = 0;
    e->timing.full_launches = 0;
}

StreamParams stream_params(const lacx_config& c, int channels) {
    StreamParams sp;
    sp.sample_rate = c.sample_rate;
    sp.bit_depth = c.bit_depth;
    sp.channels = (uint8_t)channels;
    sp.stereo_mode = channels == 2 ? c.stereo_mode : 0;
    return sp;
}

void put32(uint8_t* p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}

// Bytes reserved for the payload of a shard.  The buffer is virtual memory until touched, so the bound
// is generous: 12 bytes per sample (the costliest realistic material, full-scale 24-bit noise, needs
// about 3.3).  The real size is known from the plans before any block is published; a stream that
// exceeded the reservation (only constructible with adversarial data) is reported as a runtime error
// instead of overrunning the buffer.
uint64_t payload_upper_bound(uint64_t frames, int channels, uint32_t nb) {
    return frames * (uint64_t)channels * 12u + (uint64_t)nb * 1024u + 64u;
}

int upload(lacx_encoder* e, const int32_t* left, const int32_t* right, uint64_t frames) {
    const auto t0 = clk::now();
    int rc = ensure_pcm(e, frames, right != nullptr);
    if (rc) return rc;
    HIP_TRY(e, hipMemcpyAsync(e->d_left, left, frames * sizeof(int32_t), hipMemcpyHostToDevice, e->stream[0]),
            "H2D left");
    if (right)
        HIP_TRY(e, hipMemcpyAsync(e->d_right, right, frames * sizeof(int32_t), hipMemcpyHostToDevice, e->stream[0]),
                "H2D right");
    HIP_TRY(e, hipStreamSynchronize(e->stream[0]), "H2D synchronize");
    e->timing.h2d_ms = ms_since(t0);
    return LACX_OK;
}

int prepare(lacx_encoder* e, const void* left, uint64_t frames) {
    int rc = validate_stream_args(e, left, frames);
    if (rc) return rc;
    rc = ensure_device(e);
    if (rc) return rc;
    HIP_TRY(e, hipSetDevice(e->device), "hipSetDevice");
    return LACX_OK;
}

int fill_table(lacx_encoder* e, uint8_t* buf, uint32_t nb, const std::vector<uint64_t>& offsets) {
    put32(buf + 10, nb);
    for (uint32_t b = 0; b < nb; ++b) {
        const uint64_t size = offsets[b + 1] - offsets[b];
        if (size == 0 || size > 0xFFFFFFFFull)
            return fail(e, LACX_E_RUNTIME, "encoded block size is outside format limits");
        put32(buf + 14 + 8ull * b, e->h_bplans[b].frames);
        put32(buf + 18 + 8ull * b, (uint32_t)size);
    }
    return LACX_OK;
}

}  // namespace lacx_host

extern "C" {

int lacx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lacx_encoder_create(const lacx_config* cfg, lacx_encoder** out) {
    if (!cfg || !out) return LACX_E_INVALID;
    if (cfg->device == LACX_DEVICE_ALL) {  // every visible device (counting them does not initialise one)
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n < 1) n = 1;
        n = std::min<int>(n, (int)LACX_MAX_FANOUT);
        int32_t devs[LACX_MAX_FANOUT];
        for (int i = 0; i < n; ++i) devs[i] = i;
        return lacx_encoder_create_multi(cfg, devs, (uint32_t)n, 0, out);
    }
    lacx_encoder* e = new lacx_encoder();
    e->cfg = *cfg;
    e->knobs = read_knobs();
    *out = e;
    return LACX_OK;
}

void lacx_encoder_destroy(lacx_encoder* e) {
    if (!e) return;
    destroy_fanout(e);  // the lanes' threads, communicators and child encoders first
    e->uploader.reset();
    e->pool.reset();
    std::free(e->view_buf);
    std::free(e->view_table);
    if (e->device_ready) {
        (void)hipSetDevice(e->device);
        free_workspace(e);
        if (e->d_left) (void)hipFree(e->d_left);
        if (e->d_right) (void)hipFree(e->d_right);
        if (e->h_plans) (void)hipHostFree(e->h_plans);
        if (e->h_bplans) (void)hipHostFree(e->h_bplans);
        for (auto& row : e->ev)
            for (auto& ev : row)
                if (ev) (void)hipEventDestroy(ev);
        for (auto& ev : e->done)
            if (ev) (void)hipEventDestroy(ev);
        for (auto& ev : e->copied)
            if (ev) (void)hipEventDestroy(ev);
        if (e->prologue) (void)hipEventDestroy(e->prologue);
        for (auto& ev : e->aux_ev)
            if (ev) (void)hipEventDestroy(ev);
        if (e->pack_done) (void)hipEventDestroy(e->pack_done);
        if (e->pack_stream) (void)hipStreamDestroy(e->pack_stream);
        if (e->d_payload) (void)hipFree(e->d_payload);
        if (e->slots) (void)hipFree(e->slots);
        if (e->d_raw) (void)hipFree(e->d_raw);
        if (e->d_batch) (void)hipFree(e->d_batch);
        if (e->d_wide) (void)hipFree(e->d_wide);
        if (e->h_payload_base) (void)hipHostFree(e->h_payload_base);
        if (e->h_table) (void)hipHostFree(e->h_table);
        if (e->h_totals) (void)hipHostFree(e->h_totals);
        if (e->h_err) (void)hipHostFree(e->h_err);
        if (e->h_emitted) (void)hipHostFree(e->h_emitted);
        if (e->h_sizes) (void)hipHostFree(e->h_sizes);
        if (e->h_tspan) (void)hipHostFree(e->h_tspan);
        if (e->up_stream) (void)hipStreamDestroy(e->up_stream);
        if (e->front_stream) (void)hipStreamDestroy(e->front_stream);
        for (auto& ev : e->up_ev)
            if (ev) (void)hipEventDestroy(ev);
        if (e->h_range) (void)hipHostFree(e->h_range);
        if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
        if (e->copy_stream2) (void)hipStreamDestroy(e->copy_stream2);
        for (auto& s : e->stream)
            if (s) (void)hipStreamDestroy(s);
    }
    delete e;
}

const char* lacx_last_error(const lacx_encoder* e) { return e ? e->err.c_str() : "null encoder"; }

void lacx_free(void* p) { std::free(p); }

void lacx_get_timing(const lacx_encoder* e, lacx_timing* out) {
    if (e && out) *out = e->timing;
}

uint32_t lacx_sizeof(const char* name) {
    if (!name) return 0;
    const struct {
        const char* name;
        size_t size;
    } table[] = {{"config", sizeof(lacx_config)},           {"channel_plan", sizeof(lacx_channel_plan)},
                 {"block_plan", sizeof(lacx_block_plan)},   {"timing", sizeof(lacx_timing)},
                 {"pcm", sizeof(lacx_pcm)},                 {"batch_item", sizeof(lacx_batch_item)},
                 {"batch_out", sizeof(lacx_batch_out)},     {"wav_info", sizeof(lacx_wav_info)},
                 {"fanout_shard", sizeof(lacx_fanout_shard)}, {"fanout_out", sizeof(lacx_fanout_out)},
                 {"fanout_stats", sizeof(lacx_fanout_stats)}, {"stream_info", sizeof(lacx_stream_info)}};
    for (const auto& t : table)
        if (std::strcmp(name, t.name) == 0) return (uint32_t)t.size;
    return 0;
}

}  // extern "C"
