// lacx_api.cpp -- implementation of the C ABI declared in include/lacx.h.
//
// Thin host layer: owns the HIP streams, the device workspace and pinned plan buffers, enqueues the
// kernels (kernels.hip), and runs the host emit (emit.cpp) from the returned plan records.
// Encode calls are pipelined: the stream is cut into chunks of blocks whose kernels alternate between
// a few HIP streams; as soon as a chunk's plan records have landed in pinned host memory the emit
// workers start on its blocks while the GPU analyses the next chunk.
// There is no CPU analysis path here: without a usable HIP device every analysing call fails.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "emit.h"
#include "kernels.h"
#include "wav_parse.h"
#include "lacx.h"

using namespace lacx;

static_assert(sizeof(lacx_channel_plan) == sizeof(ChannelPlan), "ABI plan layout");
static_assert(sizeof(lacx_block_plan) == sizeof(BlockPlan), "ABI block plan layout");
static_assert(sizeof(ChannelPlan) == 296, "ChannelPlan layout");

namespace {
constexpr int kStreams = 4;
constexpr int kMaxChunks = 16;
constexpr uint32_t kMinChunkBlocks = 192;  // >= 1.5 rounds of 1024-thread workgroups over 256 CUs
using clk = std::chrono::steady_clock;
struct Chunk {
    uint32_t first, count;
};
}  // namespace

struct lacx_encoder {
    lacx_config cfg{};
    bool device_ready = false;
    int device = 0;
    hipStream_t stream[kStreams] = {};
    hipEvent_t ev[kMaxChunks][6] = {};  // [5]: after the device emit kernels
    hipEvent_t done[kMaxChunks] = {};
    hipEvent_t copied[kMaxChunks] = {};
    DeviceWorkspace ws{};
    uint32_t ws_blocks = 0;
    int32_t* d_left = nullptr;
    int32_t* d_right = nullptr;
    uint64_t d_cap = 0;
    ChannelPlan* h_plans = nullptr;  // pinned
    BlockPlan* h_bplans = nullptr;   // pinned
    uint32_t h_blocks = 0;
    // device emit
    uint8_t* d_payload = nullptr;
    uint64_t d_payload_cap = 0;
    // device-emit encode in flight between encode_device_begin and encode_device_end
    struct {
        bool active = false;
        std::vector<Chunk> chunks;
        uint32_t nb = 0;
        int channels = 0;
        bool staged = false;
        bool fused = false;
        bool drained = false;     // packer -> d_payload, copy engine -> h_payload (see h_range)
        uint32_t ranges = 0;
        hipStream_t st[4] = {};
        clk::time_point t0;
        // inputs of the call, kept for the re-emit after a too-small result reservation
        const int32_t* d_left = nullptr;
        const int32_t* d_right = nullptr;
        uint64_t frames = 0;
        int layout = 0;
    } pend;
    hipEvent_t prologue = nullptr;  // per-call memsets done (the chunk streams wait for it)
    hipStream_t pack_stream = nullptr;  // the streaming packer of the fused emit runs here, beside the analysis
    hipEvent_t pack_done = nullptr;
    uint8_t* slots = nullptr;  // staging slots of the fused emit (device memory)
    unsigned long long slots_cap = 0;
    uint8_t* d_raw = nullptr;  // WAV data chunk as read from the file (lacx_encode_wav)
    uint64_t d_raw_cap = 0;
    uint8_t* h_payload = nullptr;  // pinned: start of the payload inside h_payload_base
    uint64_t h_payload_cap = 0;
    uint8_t* h_payload_base = nullptr;  // the allocation: h_prefix bytes in front of the payload take the container's
    uint64_t h_prefix = 0;              // header + block table, so that a whole .lac can be handed out without a copy
    uint32_t* h_table = nullptr;   // pinned, [blocks][2]
    unsigned long long* h_totals = nullptr;  // pinned, per chunk payload bytes
    uint32_t* h_err = nullptr;     // pinned, per chunk
    uint32_t* h_emitted = nullptr; // pinned copy of ws.emitted (statistics of the fused emit)
    uint32_t h_emitted_cap = 0;
    unsigned long long* d_tspan = nullptr;  // [2][kMaxChunks]: ~first-start / last-end device clock of k_analyze<16,1024>
    uint8_t* zero_region = nullptr;         // one allocation for everything that is zeroed before every call
    size_t zero_bytes = 0;
    unsigned long long* h_tspan = nullptr;  // pinned copy
    uint32_t h_table_blocks = 0;
    uint8_t* view_buf = nullptr;   // result of the host-emit fallback kept alive for the *_view API
    uint32_t* view_table = nullptr;
    // lacx_encode_batch_device: the set's descriptors (host copy, device table + stream of every stream index)
    std::vector<StreamDesc> batch_streams;
    uint8_t* d_batch = nullptr;
    size_t d_batch_cap = 0;
    int32_t* d_wide = nullptr;  // lacx_block_encode outside the 25-bit domain: the eleven candidate residuals (wide.hip)
    // Copy-engine drain of the payload (one-stream encodes): the packer packs into d_payload (HBM) and reports complete
    // ranges in h_range (pinned); encode_device_end lets a copy engine fetch them while the analysis still runs.
    hipStream_t copy_stream = nullptr;
    hipStream_t copy_stream2 = nullptr;  // (ranges alternate between two streams: the next copy's set-up overlaps the current one's transfer)
    unsigned long long* h_range = nullptr;  // pinned, [h_range_cap]
    uint32_t h_range_cap = 0;
    uint32_t* d_range_cnt = nullptr;        // device, inside zero_region
    unsigned long long* d_range_end = nullptr;
    std::unique_ptr<EmitPool> pool;
    std::string err;
    lacx_timing timing{};
};

namespace {

double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

int fail(lacx_encoder* e, int code, const std::string& msg) {
    if (e) e->err = msg;
    return code;
}

int hip_fail(lacx_encoder* e, hipError_t err, const char* what) {
    return fail(e, LACX_E_DEVICE, std::string(what) + ": " + hipGetErrorString(err));
}

#define HIP_TRY(e, call, what)                                  \
    do {                                                        \
        const hipError_t _err = (call);                         \
        if (_err != hipSuccess) return hip_fail(e, _err, what); \
    } while (0)

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
        // Pipeline chunk c runs on stream c: earlier chunks get the higher priority so that they finish their
        // analysis first and their emit (PCIe-bound) runs under the later chunks' analysis.
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const bool prio = std::getenv("LACX_NO_STREAM_PRIORITY") == nullptr && greatest < least;
        int i = 0;
        for (auto& s : e->stream) {
            if (prio) {
                const int p = std::min(greatest + i, least);
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
    HIP_TRY(e, hipHostMalloc((void**)&e->h_totals, sizeof(unsigned long long) * kMaxChunks, 0), "hipHostMalloc");
    HIP_TRY(e, hipHostMalloc((void**)&e->h_err, sizeof(uint32_t) * (kMaxChunks + 4), 0), "hipHostMalloc");
    {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        HIP_TRY(e, hipStreamCreateWithPriority(&e->pack_stream, hipStreamNonBlocking, greatest), "hipStreamCreate");
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
    if (e->ws.block_off) (void)hipFree(e->ws.block_off);
    if (e->ws.table) (void)hipFree(e->ws.table);
    if (e->ws.stream_pre) (void)hipFree(e->ws.stream_pre);
    if (e->zero_region) (void)hipFree(e->zero_region);  // size_rec, ready_rec, tspan, emitted, packed, err_flag
    e->zero_region = nullptr;
    e->d_tspan = nullptr;
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
                                 sizeof(unsigned long long) * 2 * kMaxChunks + sizeof(uint32_t) * (kMaxChunks + 4) +
                                 ranges * (sizeof(unsigned long long) + sizeof(uint32_t)) + 16;
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
            p += sizeof(uint32_t) * (kMaxChunks + 4);
            p = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(p) + 7) & ~(uintptr_t)7);
            e->d_range_end = reinterpret_cast<unsigned long long*>(p);
            p += ranges * sizeof(unsigned long long);
            e->d_range_cnt = reinterpret_cast<uint32_t*>(p);
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
int ensure_slots(lacx_encoder* e, uint32_t nblocks, int channels, int bit_depth = 0) {
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
                          int layout = 0) {
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
    const char* dbg = std::getenv("LACX_DEBUG_SKIP");  // timing ablations only
    prm.debug_skip = dbg ? (uint32_t)std::strtoul(dbg, nullptr, 0) : 0u;
    return prm;
}

// Launch set of one stream (or one pipeline chunk of it): the descriptor travels in the kernel arguments.
LaunchSet one_stream_set(const AnalyzeParams& prm, const int32_t* left, const int32_t* right, uint32_t fuse_items = 0,
                         uint64_t out_cap = 0) {
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
    w.table += (size_t)first_block * 2;
    return w;
}

// Host emit wants many chunks (emit of chunk i overlaps the analysis of chunk i+1); with the emit on the
// device the only host work is a copy, and two chunks (payload copy of one under the kernels of the other)
// measured best.
std::vector<Chunk> plan_chunks(uint32_t nb, bool device_emit = false, bool fused = false, bool upload = false) {
    uint32_t nchunks = nb / kMinChunkBlocks;
    // device emit without the fused path: 3 chunks up to an hour of stereo 48 kHz per call, 4 and 6 beyond (measured on a
    // 2 h shard).  With the fused emit + streaming packer nothing is left to overlap by chunking -- the payload leaves
    // while the analysis runs, and ingest / probes keep every CU busy by themselves -- and one launch set measured best
    // from 10 min to 2 h of audio (a chunked run only adds kernel boundaries).
    // With the input still in host memory the chunks pipeline the upload: three chunks of relative size 1 : 3 : 4 -- a
    // small first one, so that little of the H2D copy is exposed before the first kernel (measured, 10 min stream, WAV
    // image -> .lac: 4 equal chunks 4.36 ms, 1:2:3:3 4.17, 1:2:3 4.17, 1:3:4 4.11, 2:3:4 4.21, 6 or 8 equal 4.6).
    const uint32_t dev_chunks = fused ? (upload ? 3u : 1u) : (nb >= 12000u ? 6u : (nb >= 6000u ? 4u : 3u));
    nchunks = std::max(1u, std::min(nchunks, device_emit ? dev_chunks : 8u));
    bool forced = false;
    if (const char* env = std::getenv("LACX_PIPE_CHUNKS")) {  // tuning knob
        const unsigned long v = std::strtoul(env, nullptr, 0);
        if (v >= 1 && v <= (unsigned long)kMaxChunks) {
            nchunks = std::min<uint32_t>((uint32_t)v, nb);
            forced = true;
        }
    }
    std::vector<Chunk> out;
    const char* split_env = std::getenv("LACX_PIPE_SPLIT");  // tuning knob: relative chunk sizes, e.g. "5,3,1"
    // Device emit: three chunks on three streams of falling priority, the last one a little smaller -- its
    // emit is the only one whose PCIe writes are not hidden under another chunk's analysis (measured best).
    if (!split_env && !forced && device_emit && nchunks == 3u) split_env = (fused && upload) ? "1,3,4" : "5,5,4";
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

// Enqueues the kernels + plan D2H of one chunk on stream `st`, then records done[c].
int enqueue_chunk(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, uint64_t frames, int channels,
                  int stereo_mode, int bit_depth, const Chunk& ck, int c, hipStream_t st) {
    const uint64_t f0 = (uint64_t)ck.first * kMaxBlock;
    const uint64_t f1 = std::min<uint64_t>(frames, (uint64_t)(ck.first + ck.count) * kMaxBlock);
    const AnalyzeParams prm = make_params(e, f1 - f0, channels, stereo_mode, bit_depth);
    const DeviceWorkspace w = ws_at(e->ws, ck.first);
    LaunchSet ls = one_stream_set(prm, d_left + f0, d_right ? d_right + f0 : nullptr);
    HIP_TRY(e, launch_analysis(bind(ls), w, st, e->ev[c]), "kernel launch");
    HIP_TRY(e, hipMemcpyAsync(e->h_plans + (size_t)ck.first * kSlotsPerBlock, w.plans,
                              (size_t)ck.count * kSlotsPerBlock * sizeof(ChannelPlan), hipMemcpyDeviceToHost, st),
            "D2H plans");
    HIP_TRY(e, hipMemcpyAsync(e->h_bplans + ck.first, w.bplans, (size_t)ck.count * sizeof(BlockPlan),
                              hipMemcpyDeviceToHost, st),
            "D2H block plans");
    HIP_TRY(e, hipEventRecord(e->done[c], st), "event record");
    return LACX_OK;
}

// Sample-range errors in the reference's order: all of left first, then right (ref lac/encoder.cpp:238-241).
int check_sample_range(lacx_encoder* e, uint32_t nb) {
    for (int pass = 0; pass < 2; ++pass) {
        for (uint32_t b = 0; b < nb; ++b) {
            const BlockPlan& bp = e->h_bplans[b];
            if (!bp.invalid) continue;
            const bool is_right = (bp.first_bad >> 31) != 0;
            // per block the left channel wins the minimum, so a "right" entry means a clean left channel
            if ((pass == 0) == is_right) continue;
            const uint64_t idx = (uint64_t)b * kMaxBlock + (bp.first_bad & 0x7FFFFFFFu);
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

// Runs the kernels on device-resident PCM in one launch set on `st`; leaves plans in the pinned buffers.
int analyze_on_device(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, uint64_t frames,
                      int channels, int stereo_mode, int bit_depth, hipStream_t st) {
    const uint32_t nb = blocks_for(frames);
    int rc = ensure_workspace(e, nb);
    if (rc) return rc;
    const auto t0 = clk::now();
    const Chunk all{0, nb};
    rc = enqueue_chunk(e, d_left, d_right, frames, channels, stereo_mode, bit_depth, all, 0, st);
    if (rc) return rc;
    HIP_TRY(e, hipEventSynchronize(e->done[0]), "event synchronize");
    e->timing.d2h_ms = ms_since(t0);
    reset_device_timing(e);
    add_chunk_timing(e, 0);
    e->timing.full_launches = 1;
    count_slots(e, 0, nb);
    return LACX_OK;
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

// Pipelined analysis + emit.  `head` bytes are reserved in front of the payload (container header +
// table for whole-stream calls, 0 for shards).  On success *buf_out holds head + payload (malloc'd).
int encode_pipelined(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, const int32_t* h_left,
                     const int32_t* h_right, uint64_t frames, hipStream_t user_stream, uint64_t head,
                     uint8_t** buf_out, uint64_t* payload_size, std::vector<uint64_t>& offsets) {
    const int channels = d_right ? 2 : 1;
    const uint32_t nb = blocks_for(frames);
    int rc = ensure_workspace(e, nb);
    if (rc) return rc;
    const StreamParams sp = stream_params(e->cfg, channels);
    const std::vector<Chunk> chunks = plan_chunks(nb);
    const uint64_t cap = payload_upper_bound(frames, channels, nb);
    uint8_t* buf = static_cast<uint8_t*>(std::malloc(head + cap));
    if (!buf) return fail(e, LACX_E_RUNTIME, "out of memory");
    offsets.assign((size_t)nb + 1, 0);

    reset_device_timing(e);
    const auto t0 = clk::now();
    // the caller's stream (if any) carries chunks 0, 2, ...; the encoder's second stream the others
    hipStream_t st[kStreams];
    for (int i = 0; i < kStreams; ++i) st[i] = e->stream[i];
    if (user_stream) st[0] = user_stream;
    // Work the caller queued on its stream (e.g. the kernel or copy that produces the PCM) must be ordered before
    // every chunk, also those that run on the encoder's own streams: they wait for an event recorded on st[0].
    {
        const hipError_t pe = hipEventRecord(e->prologue, st[0]);
        if (pe != hipSuccess) {
            std::free(buf);
            return hip_fail(e, pe, "event record");
        }
    }
    for (size_t c = 0; c < chunks.size(); ++c) {
        if (c % kStreams != 0) {
            const hipError_t we = hipStreamWaitEvent(st[c % kStreams], e->prologue, 0);
            if (we != hipSuccess) {
                (void)hipDeviceSynchronize();
                std::free(buf);
                return hip_fail(e, we, "stream wait");
            }
        }
        rc = enqueue_chunk(e, d_left, d_right, frames, channels, e->cfg.stereo_mode, e->cfg.bit_depth, chunks[c],
                           (int)c, st[c % kStreams]);
        if (rc) {
            (void)hipDeviceSynchronize();
            std::free(buf);
            return rc;
        }
    }
    EmitPool& pool = pool_of(e);
    pool.begin(sp, h_left, h_right, frames, e->h_bplans, e->h_plans, nb, offsets.data(), buf + head);
    uint64_t off = 0;
    int status = LACX_OK;
    for (size_t c = 0; c < chunks.size(); ++c) {
        const hipError_t he = hipEventSynchronize(e->done[c]);
        if (he != hipSuccess) {
            status = hip_fail(e, he, "event synchronize");
            break;
        }
        const Chunk& ck = chunks[c];
        bool bad = false;
        for (uint32_t b = ck.first; b < ck.first + ck.count; ++b) {
            if (e->h_bplans[b].invalid) bad = true;
            offsets[b] = off;
            off += block_payload_bytes(sp, e->h_bplans[b], e->h_plans + (size_t)b * kSlotsPerBlock);
        }
        offsets[ck.first + ck.count] = off;
        if (bad || off > cap) {
            status = bad ? LACX_E_INVALID : fail(e, LACX_E_RUNTIME, "payload exceeds the reserved bound");
            break;
        }
        pool.publish(ck.first + ck.count);
    }
    e->timing.d2h_ms = ms_since(t0);
    if (status != LACX_OK) {
        pool.abort();
        (void)pool.finish();
        (void)hipDeviceSynchronize();
        std::free(buf);
        if (status == LACX_E_INVALID) {
            const int rr = check_sample_range(e, nb);  // formats the reference's message
            return rr ? rr : fail(e, LACX_E_INVALID, "sample outside the configured PCM bit depth");
        }
        return status;
    }
    const bool ok = pool.finish();
    for (size_t c = 0; c < chunks.size(); ++c) {
        add_chunk_timing(e, (int)c);
        count_slots(e, chunks[c].first, chunks[c].count);
    }
    e->timing.full_launches = (uint32_t)chunks.size();
    if (!ok) {
        std::free(buf);
        return fail(e, LACX_E_RUNTIME, "emitted size disagrees with the device plan (internal error)");
    }
    uint8_t* shrunk = static_cast<uint8_t*>(std::realloc(buf, (head + off) ? (head + off) : 1));
    *buf_out = shrunk ? shrunk : buf;
    *payload_size = off;
    return LACX_OK;
}

// Device-emit pipeline: per chunk the kernels also produce the bitstream (k_offsets + k_emit), written by the
// kernel straight into one pinned host buffer at global byte offsets (chunk c starts where chunk c-1 ends).
// Results stay in encoder-owned pinned memory (e->h_payload, e->h_table).  Returns LACX_OK, an error, or -1 when the reservation of the
// pinned buffer was too small (the caller then falls back to the host-emit pipeline, same bytes).
// Kernel arguments of pipeline chunk c of a device-emit encode.
struct ChunkCtx {
    AnalyzeParams prm;
    const int32_t* left;
    const int32_t* right;
    DeviceWorkspace w;
    // the chunk as a launch set of one stream; fuse_items: its stream indices that take part in the fused emit,
    // out_cap: capacity of the result buffer (offsets are shard-wide: out_base 0)
    LaunchSet set(uint32_t shard_fuse_items, uint64_t out_cap) const {
        const uint32_t items = prm.num_blocks * (uint32_t)prm.channels;
        const uint32_t mine = shard_fuse_items > prm.stream_base ? std::min(items, shard_fuse_items - prm.stream_base) : 0u;
        return one_stream_set(prm, left, right, mine, out_cap);
    }
};
ChunkCtx chunk_ctx(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, uint64_t frames, int layout,
                   int channels, const Chunk& ck, size_t c) {
    ChunkCtx x;
    const uint64_t frame_bytes = layout == 1 ? 2ull * channels : (layout == 2 ? 3ull * channels : 4ull);
    const uint64_t f0 = (uint64_t)ck.first * kMaxBlock;
    const uint64_t f1 = std::min<uint64_t>(frames, (uint64_t)(ck.first + ck.count) * kMaxBlock);
    x.prm = make_params(e, f1 - f0, channels, e->cfg.stereo_mode, e->cfg.bit_depth, layout);
    x.prm.stream_base = ck.first * (uint32_t)channels;
    // chunk base pointers: planar int32 advances by frames, interleaved layouts by bytes
    x.left = layout ? reinterpret_cast<const int32_t*>(reinterpret_cast<const uint8_t*>(d_left) + f0 * frame_bytes)
                    : d_left + f0;
    x.right = (!layout && d_right) ? d_right + f0 : nullptr;
    x.w = ws_at(e->ws, ck.first);
    x.w.block_off = e->ws.block_off + ck.first + c;  // count + 1 entries per chunk
    x.w.err_flag = e->ws.err_flag + c;
    x.w.t_first = e->d_tspan + c;
    x.w.t_last = e->d_tspan + kMaxChunks + c;
    return x;
}

// Size of the pinned result reservation: 1.25 x the PCM at its source bit depth covers every realistic stream (the
// exact size is only known after the analysis; a stream that needs more is re-emitted into a regrown buffer, see
// reemit_into_regrown_buffer).  LACX_PINNED_CAP_BYTES overrides the estimate (tests force the regrow path with it).
uint64_t pinned_reservation(const lacx_encoder* e, uint64_t frames, int channels, uint32_t nb) {
    if (const char* env = std::getenv("LACX_PINNED_CAP_BYTES")) {
        const unsigned long long v = std::strtoull(env, nullptr, 0);
        if (v > 0) return (uint64_t)v;
    }
    return frames * (uint64_t)channels * (e->cfg.bit_depth / 8u) * 5u / 4u + (uint64_t)nb * 64u + 4096u;
}

// Host-resident input of an encode whose upload is pipelined with the analysis: chunk c's PCM is copied to the device
// on chunk c's stream right in front of its kernels, so the upload of chunk c+1 overlaps the analysis of chunk c
// (ref src/main.cpp:658-675 reads the whole file first, then encodes).
struct HostSrc {
    const uint8_t* p0 = nullptr;  // planar: left; interleaved: the WAV data chunk
    const uint8_t* p1 = nullptr;  // planar: right (null for mono)
    uint64_t frame_bytes = 0;     // bytes per frame in p0 (and p1)
};

// Part 1: enqueue everything (no host synchronisation).
int encode_device_begin_impl(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, uint64_t frames,
                             hipStream_t user_stream, int layout, int layout_channels, const HostSrc* hs);
int encode_device_begin(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, uint64_t frames,
                        hipStream_t user_stream, int layout = 0, int layout_channels = 0, const HostSrc* hs = nullptr) {
    const int rc = encode_device_begin_impl(e, d_left, d_right, frames, user_stream, layout, layout_channels, hs);
    // A failure half-way leaves kernels queued that write to the workspace, the slots and the pinned result buffer: they
    // must have drained before the next call clears, frees or regrows any of those.
    if (rc != LACX_OK && e->device_ready) (void)hipDeviceSynchronize();
    return rc;
}
int encode_device_begin_impl(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, uint64_t frames,
                             hipStream_t user_stream, int layout, int layout_channels, const HostSrc* hs) {
    if (e->pend.active) return fail(e, LACX_E_RUNTIME, "an encode is already in flight on this encoder");
    const int channels = layout ? layout_channels : (d_right ? 2 : 1);
    const uint32_t nb = blocks_for(frames);
    int rc = ensure_workspace(e, nb);
    if (rc) return rc;
    // Emit fused into the analysis kernel (default; LACX_FUSED_EMIT=0 leaves the bitstream to k_offsets + k_emit alone;
    // k_emit runs after the analysis in any case and picks up whatever the fused path did not write).
    const char* fenv = std::getenv("LACX_FUSED_EMIT");
    const bool fused = !(fenv && *fenv == '0');
    e->pend.chunks = plan_chunks(nb, true, fused, hs != nullptr);
    const std::vector<Chunk>& chunks = e->pend.chunks;
    // Destination of k_emit: by default the pinned host buffer itself (the kernel's 16-byte stores cross PCIe
    // while later blocks are still being analysed, so no separate D2H pass is left at the end); with
    // LACX_EMIT_STAGED=1 a device arena sized for the worst case (12 bytes per sample), copied afterwards.
    static const bool staged = [] {
        const char* v = std::getenv("LACX_EMIT_STAGED");
        return v && *v && *v != '0';
    }();
    // Default with the fused emit: the packer packs into device memory and a copy engine drains it (LACX_DIRECT_PACKER=1:
    // the packer's CUs store straight into pinned host memory, the round-2 layout).
    const bool drained = !staged && !(std::getenv("LACX_FUSED_EMIT") && *std::getenv("LACX_FUSED_EMIT") == '0') &&
                         !std::getenv("LACX_DIRECT_PACKER") && !std::getenv("LACX_NO_PACKER") && !std::getenv("LACX_PINNED_CAP_BYTES");
    if (drained) {
        const uint64_t dev_cap = pinned_reservation(e, frames, channels, nb) + 64ull;
        if (dev_cap > e->d_payload_cap) {
            if (e->d_payload) (void)hipFree(e->d_payload);
            e->d_payload = nullptr;
            e->d_payload_cap = 0;
            HIP_TRY(e, hipMalloc((void**)&e->d_payload, dev_cap), "hipMalloc(payload)");
            e->d_payload_cap = dev_cap;
        }
        const uint32_t ranges = (nb * (uint32_t)channels + kPackerRangeItems - 1u) / kPackerRangeItems + 1u;
        if (ranges > e->h_range_cap) {
            if (e->h_range) (void)hipHostFree(e->h_range);
            e->h_range = nullptr;
            e->h_range_cap = 0;
            HIP_TRY(e, hipHostMalloc((void**)&e->h_range, (size_t)ranges * sizeof(unsigned long long), 0), "hipHostMalloc(ranges)");
            e->h_range_cap = ranges;
        }
    }
    if (staged) {
        const uint64_t dev_cap = payload_upper_bound(frames, channels, nb) + 64ull;
        if (dev_cap > e->d_payload_cap) {
            if (e->d_payload) (void)hipFree(e->d_payload);
            e->d_payload = nullptr;
            e->d_payload_cap = 0;
            HIP_TRY(e, hipMalloc((void**)&e->d_payload, dev_cap), "hipMalloc(payload)");
            e->d_payload_cap = dev_cap;
        }
    }
    const uint64_t host_cap = pinned_reservation(e, frames, channels, nb);
    const uint64_t prefix = (14ull + 8ull * nb + 4095ull) & ~4095ull;  // room for the container header + block table
    if (host_cap > e->h_payload_cap || prefix > e->h_prefix || std::getenv("LACX_PINNED_CAP_BYTES")) {
        if (e->h_payload_base) (void)hipHostFree(e->h_payload_base);
        e->h_payload = e->h_payload_base = nullptr;
        e->h_payload_cap = e->h_prefix = 0;
        HIP_TRY(e, hipHostMalloc((void**)&e->h_payload_base, prefix + host_cap, 0), "hipHostMalloc(payload)");
        e->h_payload = e->h_payload_base + prefix;
        e->h_payload_cap = host_cap;
        e->h_prefix = prefix;
    }
    if (nb > e->h_table_blocks) {
        if (e->h_table) (void)hipHostFree(e->h_table);
        e->h_table = nullptr;
        HIP_TRY(e, hipHostMalloc((void**)&e->h_table, (size_t)nb * 2 * sizeof(uint32_t), 0), "hipHostMalloc(table)");
        e->h_table_blocks = nb;
    }
    reset_device_timing(e);
    e->timing.emit_ms = 0;
    const auto t0 = clk::now();
    for (int c = 0; c < kMaxChunks; ++c) e->h_totals[c] = 0;  // (the last chunk's total doubles as the "all kernels done" word)
    hipStream_t st[kStreams];
    for (int i = 0; i < kStreams; ++i) st[i] = e->stream[i];
    if (user_stream) st[0] = user_stream;
    uint8_t* emit_dst = e->d_payload;
    uint64_t emit_cap = e->d_payload_cap;
    if (drained) {
        emit_cap = std::min<uint64_t>(e->d_payload_cap, e->h_payload_cap);
    } else if (!staged) {
        HIP_TRY(e, hipHostGetDevicePointer((void**)&emit_dst, e->h_payload, 0), "hipHostGetDevicePointer");
        emit_cap = e->h_payload_cap;
    }
    const unsigned long long* prev_end = nullptr;  // device address of the byte total of the chunks so far
    HIP_TRY(e, hipMemsetAsync(e->zero_region, 0, e->zero_bytes, st[0]), "memset");  // records, flags, time stamps
    if (fused) {
        rc = ensure_slots(e, nb, channels);
        if (rc) return rc;
        if (nb * 2u > e->h_emitted_cap) {
            if (e->h_emitted) (void)hipHostFree(e->h_emitted);
            e->h_emitted = nullptr;
            e->h_emitted_cap = 0;
            HIP_TRY(e, hipHostMalloc((void**)&e->h_emitted, (size_t)nb * 2 * sizeof(uint32_t), 0), "hipHostMalloc(emitted)");
            e->h_emitted_cap = nb * 2u;
        }
    } else {
        e->ws.slots = nullptr;
    }
    HIP_TRY(e, hipEventRecord(e->prologue, st[0]), "event record");
    // Stream indices that take part in the fused emit: all but those of a final block of <= 4096 frames in per-block
    // stereo mode, which may be encoded both ways and compared afterwards (ref lac/encoder.cpp:336-340).
    uint32_t fuse_items = 0;
    if (fused) {
        const uint64_t last_frames = frames - (uint64_t)(nb - 1) * kMaxBlock;
        const bool last_both_ways = channels == 2 && e->cfg.stereo_mode == 2 && last_frames <= (uint64_t)kFullCompareLimit;
        fuse_items = (nb - (last_both_ways ? 1u : 0u)) * (uint32_t)channels;
    }
    for (size_t c = 0; c < chunks.size(); ++c) {
        const Chunk& ck = chunks[c];
        hipStream_t s = st[c % kStreams];
        if (c % kStreams != 0) HIP_TRY(e, hipStreamWaitEvent(s, e->prologue, 0), "stream wait");
        const ChunkCtx cx = chunk_ctx(e, d_left, d_right, frames, layout, channels, ck, c);
        const AnalyzeParams& prm = cx.prm;
        const int32_t *cl = cx.left, *cr = cx.right;
        const DeviceWorkspace& w = cx.w;
        FuseArgs fa;
        if (fused) {
            fa.slots = e->ws.slots;
            fa.slot_stride = e->ws.slot_stride;
            fa.emitted = e->ws.emitted;
            fa.err_flag = w.err_flag;
            fa.size_rec = e->ws.size_rec;
            fa.ready_rec = e->ws.ready_rec;
        }
        if (hs) {  // this chunk's PCM: host -> device, on the chunk's stream, right in front of its kernels
            const auto th0 = clk::now();
            const uint64_t f0 = (uint64_t)ck.first * kMaxBlock;
            const uint64_t f1 = std::min<uint64_t>(frames, (uint64_t)(ck.first + ck.count) * kMaxBlock);
            const uint64_t o = f0 * hs->frame_bytes, nbytes = (f1 - f0) * hs->frame_bytes;
            HIP_TRY(e, hipMemcpyAsync(const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(d_left)) + o, hs->p0 + o, nbytes,
                                      hipMemcpyHostToDevice, s), "H2D pcm");
            if (hs->p1)
                HIP_TRY(e, hipMemcpyAsync(const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(d_right)) + o, hs->p1 + o, nbytes,
                                          hipMemcpyHostToDevice, s), "H2D pcm");
            e->timing.h2d_ms += ms_since(th0);
        }
        // Fused emit: the packer walks the stream indices in order, so the whole-block kernels of the chunks run in that
        // order too (chunk c's waits for chunk c-1's: ev[c-1][3] is recorded behind it); what comes before them --
        // ingest, Levinson, probes -- still overlaps the previous chunk's analysis.
        const bool chain = fused && c > 0 && !std::getenv("LACX_NO_CHAIN");
        (void)cl;
        (void)cr;
        (void)prm;
        LaunchSet ls = cx.set(fuse_items, emit_cap);
        HIP_TRY(e, launch_analysis(bind(ls), w, s, e->ev[c], &fa, chain ? e->ev[c - 1][3] : nullptr), "kernel launch");
        if (c == 0 && fuse_items && !std::getenv("LACX_NO_PACKER")) {
            // the streaming packer: beside the whole-block analysis kernels, on its own stream.  It starts when the first
            // chunk's ingest / Levinson / probe kernels are done (ev[0][2] is recorded right in front of the whole-block
            // kernel), so its bounded waits only ever cover the progress of the analysis itself, however long the shard.
            HIP_TRY(e, hipStreamWaitEvent(e->pack_stream, e->ev[0][2], 0), "stream wait");
            // (the packer walks the whole shard: one stream whose indices start at 0)
            AnalyzeParams shard_prm = make_params(e, frames, channels, e->cfg.stereo_mode, e->cfg.bit_depth, layout);
            shard_prm.stream_base = 0;
            LaunchSet shard = one_stream_set(shard_prm, nullptr, nullptr, fuse_items, emit_cap);
            RangeProgress rp;
            if (drained) {
                const uint32_t ranges = (fuse_items + kPackerRangeItems - 1u) / kPackerRangeItems;
                std::memset(e->h_range, 0, (size_t)ranges * sizeof(unsigned long long));
                // (the range counters live in the region the call's one memset clears: a memset on the packer's own stream
                // would make the packer's dispatch wait for everything queued before it, the analysis kernel included)
                rp.range_cnt = e->d_range_cnt;
                rp.range_end = e->d_range_end;
                HIP_TRY(e, hipHostGetDevicePointer((void**)&rp.host_end, e->h_range, 0), "hipHostGetDevicePointer");
                rp.fuse_total = fuse_items;
                if (const char* fm = std::getenv("LACX_DRAIN_FENCE")) rp.fence_mode = (uint32_t)std::atoi(fm);
                e->pend.ranges = ranges;
            }
            HIP_TRY(e, launch_stream_out(bind(shard), e->ws, emit_dst, e->ws.err_flag + kMaxChunks, e->pack_stream, rp), "packer launch");
            HIP_TRY(e, hipEventRecord(e->pack_done, e->pack_stream), "event record");
        }
    }
    // Second pass over the chunks: everything behind the analysis.  With the fused emit it waits for the packer (what
    // k_pack / k_emit still have to move is only known once the packer has finished), and a stream may carry several
    // chunks, so none of this may be enqueued before the last chunk's analysis kernels.
    for (size_t c = 0; c < chunks.size(); ++c) {
        const Chunk& ck = chunks[c];
        hipStream_t s = st[c % kStreams];
        const ChunkCtx cx = chunk_ctx(e, d_left, d_right, frames, layout, channels, ck, c);
        const AnalyzeParams& prm = cx.prm;
        const int32_t *cl = cx.left, *cr = cx.right;
        const DeviceWorkspace& w = cx.w;
        // block offsets are global: chunk c starts where chunk c-1 ended (its k_offsets must have run)
        const bool packer_counts = fuse_items && !std::getenv("LACX_NO_PACKER");
        (void)cl;
        (void)cr;
        (void)prm;
        LaunchSet ls = cx.set(fuse_items, emit_cap);
        HIP_TRY(e, launch_emit(bind(ls), w, emit_dst, prev_end, c ? e->copied[c - 1] : nullptr,
                               e->copied[c], s, true, packer_counts ? e->ws.err_flag + kMaxChunks + 1 : nullptr,
                               nb * (uint32_t)channels, packer_counts ? e->pack_done : nullptr,
                               e->ws.err_flag + kMaxChunks + 3), "emit launch");
        prev_end = w.block_off + ck.count;
        HIP_TRY(e, hipEventRecord(e->ev[c][5], s), "event record");
        // what the host reads afterwards, in one kernel that stores into the pinned buffers
        GatherList g;
        auto mapped = [](auto* host) -> decltype(host) {  // the address the device uses for a pinned host buffer
            void* d = nullptr;
            return hipHostGetDevicePointer(&d, host, 0) == hipSuccess ? static_cast<decltype(host)>(d) : nullptr;
        };
        BlockPlan* m_bplans = mapped(e->h_bplans);
        uint32_t *m_table = mapped(e->h_table), *m_err = mapped(e->h_err), *m_emitted = fused ? mapped(e->h_emitted) : nullptr;
        unsigned long long *m_totals = mapped(e->h_totals), *m_tspan = mapped(e->h_tspan);
        if (!m_bplans || !m_table || !m_err || !m_totals || !m_tspan || (fused && !m_emitted))
            return fail(e, LACX_E_RUNTIME, "hipHostGetDevicePointer failed");
        g.add(w.bplans, m_bplans + ck.first, (size_t)ck.count * sizeof(BlockPlan));
        g.add(w.table, m_table + (size_t)ck.first * 2, (size_t)ck.count * 2 * sizeof(uint32_t));
        g.add(w.block_off + ck.count, &m_totals[c], sizeof(unsigned long long));
        g.add(w.err_flag, &m_err[c], sizeof(uint32_t));
        // the packer's error flags, moved count, waves that gave up, and k_pack's repacked count
        if (c + 1 == chunks.size()) g.add(e->ws.err_flag + kMaxChunks, &m_err[kMaxChunks], 4 * sizeof(uint32_t));
        if (fused)
            g.add(e->ws.packed + (size_t)ck.first * channels, m_emitted + (size_t)ck.first * channels,
                  (size_t)ck.count * channels * sizeof(uint32_t));
        g.add(w.t_first, &m_tspan[c], sizeof(unsigned long long));
        g.add(w.t_last, &m_tspan[kMaxChunks + c], sizeof(unsigned long long));
        HIP_TRY(e, launch_gather(g, s), "gather launch");
        HIP_TRY(e, hipEventRecord(e->done[c], s), "event record");
    }
    e->pend.active = true;
    e->pend.nb = nb;
    e->pend.channels = channels;
    e->pend.staged = staged;
    e->pend.fused = fused;
    e->pend.drained = drained;  // (the result is fetched from the device payload even when no range was ever reported)
    if (!(drained && fused && fuse_items != 0)) e->pend.ranges = 0;
    for (int i = 0; i < kStreams; ++i) e->pend.st[i] = st[i];
    e->pend.t0 = t0;
    e->pend.d_left = d_left;
    e->pend.d_right = d_right;
    e->pend.frames = frames;
    e->pend.layout = layout;
    return LACX_OK;
}

// The pinned result buffer was reserved from an estimate and the stream needs more.  Every chunk's k_offsets has run
// (offsets do not depend on the capacity) and the blocks that did not fit wrote nothing, so the exact total is known:
// regrow the buffer and run only the emit kernels again, chunk by chunk, from the plans still in the workspace.
int reemit_into_regrown_buffer(lacx_encoder* e, uint64_t* payload_size) {
    const std::vector<Chunk>& chunks = e->pend.chunks;
    HIP_TRY(e, hipDeviceSynchronize(), "synchronize");
    const uint64_t total = e->h_totals[chunks.size() - 1];  // cumulative byte count after the last chunk
    const uint64_t prefix = e->h_prefix;
    if (e->h_payload_base) (void)hipHostFree(e->h_payload_base);
    e->h_payload = e->h_payload_base = nullptr;
    e->h_payload_cap = e->h_prefix = 0;
    const uint64_t cap = total + 4096u;
    HIP_TRY(e, hipHostMalloc((void**)&e->h_payload_base, prefix + cap, 0), "hipHostMalloc(payload regrow)");
    e->h_payload = e->h_payload_base + prefix;
    e->h_payload_cap = cap;
    e->h_prefix = prefix;
    uint8_t* dst = nullptr;
    HIP_TRY(e, hipHostGetDevicePointer((void**)&dst, e->h_payload, 0), "hipHostGetDevicePointer");
    hipStream_t s = e->stream[0];
    HIP_TRY(e, hipMemsetAsync(e->ws.err_flag, 0, sizeof(uint32_t) * (kMaxChunks + 1), s), "memset");
    const unsigned long long* prev_end = nullptr;
    for (size_t c = 0; c < chunks.size(); ++c) {
        const ChunkCtx cx = chunk_ctx(e, e->pend.d_left, e->pend.d_right, e->pend.frames, e->pend.layout,
                                      e->pend.channels, chunks[c], c);
        LaunchSet ls = cx.set(0, cap);
        HIP_TRY(e, launch_emit(bind(ls), cx.w, dst, prev_end, nullptr, nullptr, s, /*skip_emitted=*/false), "emit relaunch");
        prev_end = cx.w.block_off + chunks[c].count;
        HIP_TRY(e, hipMemcpyAsync(&e->h_err[c], cx.w.err_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, s), "D2H err");
    }
    HIP_TRY(e, hipStreamSynchronize(s), "synchronize");
    for (size_t c = 0; c < chunks.size(); ++c) {
        if (e->h_err[c] & 1u) return fail(e, LACX_E_RUNTIME, "device emit disagrees with the analysis plan (internal error)");
        if ((e->h_err[c] & 2u) || e->h_totals[c] > cap)
            return fail(e, LACX_E_RUNTIME, "payload exceeds the regrown result buffer (internal error)");
    }
    *payload_size = total;
    return LACX_OK;
}

// Part 2: wait for the chunks in order, check them, hand the result over.  Returns LACX_OK, an error, or -1 when
// the reservation of the pinned buffer was too small.
int encode_device_end(lacx_encoder* e, uint64_t* payload_size) {
    if (!e->pend.active) return fail(e, LACX_E_RUNTIME, "no encode in flight on this encoder");
    e->pend.active = false;
    const std::vector<Chunk>& chunks = e->pend.chunks;
    const uint32_t nb = e->pend.nb;
    const int channels = e->pend.channels;
    const bool staged = e->pend.staged;
    hipStream_t* st = e->pend.st;
    const auto t0 = e->pend.t0;
    uint64_t off = 0;
    int status = LACX_OK;
    size_t copies = 0;
    // Copy-engine drain: while the kernels run, every range of stream indices the packer reports complete is fetched
    // from the device payload into the pinned result buffer (hipMemcpyAsync on its own stream: a copy engine, not CUs).
    uint64_t drained_to = 0;
    static const bool dbg_drain = std::getenv("LACX_DEBUG_DRAIN") != nullptr;
    static const bool two_streams = std::getenv("LACX_ONE_COPY_STREAM") == nullptr;
    if (e->pend.drained) {
        uint32_t next = 0;
        const volatile unsigned long long* flags = e->h_range;
        auto pump = [&]() {
            while (next < e->pend.ranges) {
                const unsigned long long v = flags[next];
                if (v == 0) break;
                const uint64_t end = v - 1u;
                if (dbg_drain) std::fprintf(stderr, "[drain] range %u end %llu at %.3f ms\n", next, (unsigned long long)end, ms_since(t0));
                if (end > drained_to && end <= e->h_payload_cap) {
                    if (hipMemcpyAsync(e->h_payload + drained_to, e->d_payload + drained_to, end - drained_to, hipMemcpyDeviceToHost,
                                       (next & 1u) && two_streams ? e->copy_stream2 : e->copy_stream) != hipSuccess)
                        return;  // (the final copy below fetches what is missing)
                    drained_to = end;
                }
                ++next;
            }
        };
        // (no runtime call in the loop but the copies: the gather kernel -- the last one of the call -- stores the
        // cumulative byte count of the last chunk, non-zero, into pinned memory that was zeroed before the launch)
        const volatile unsigned long long* finished = &e->h_totals[chunks.size() - 1];
        const auto poll0 = clk::now();
        while (*finished == 0ull) {
            pump();
            if (ms_since(poll0) > 20000.0) break;  // (a lost device: the event wait below reports it)
        }
        if (dbg_drain) std::fprintf(stderr, "[drain] kernels done at %.3f ms, copy stream %s\n", ms_since(t0),
                                    hipStreamQuery(e->copy_stream) == hipSuccess ? "idle" : "busy");
        pump();
    }
    for (size_t c = 0; c < chunks.size(); ++c) {
        const hipError_t he = hipEventSynchronize(e->done[c]);
        if (he != hipSuccess) {
            status = hip_fail(e, he, "event synchronize");
            break;
        }
        const Chunk& ck = chunks[c];
        bool bad = false;
        for (uint32_t b = ck.first; b < ck.first + ck.count; ++b) bad = bad || e->h_bplans[b].invalid;
        if (bad) {
            status = LACX_E_INVALID;
            break;
        }
        if (e->h_err[c] & 1u) {
            status = fail(e, LACX_E_RUNTIME, "device emit disagrees with the analysis plan (internal error)");
            break;
        }
        const uint64_t end = e->h_totals[c];  // cumulative
        const bool packer_overflow = c + 1 == chunks.size() && (e->h_err[kMaxChunks] & 2u);
        if ((e->h_err[c] & 2u) || packer_overflow || end > e->h_payload_cap) {
            status = -1;  // reservation too small: re-emit into a regrown buffer below
            break;
        }
        if (staged) {
            hipStream_t s = st[c % kStreams];
            const hipError_t ce = hipMemcpyAsync(e->h_payload + off, e->d_payload + off, end - off, hipMemcpyDeviceToHost, s);
            if (ce != hipSuccess || hipEventRecord(e->done[c], s) != hipSuccess) {
                status = hip_fail(e, ce, "D2H payload");
                break;
            }
            ++copies;
        }
        off = end;
    }
    for (size_t c = 0; c < copies; ++c) (void)hipEventSynchronize(e->done[c]);
    if (e->pend.drained) {
        if (status == LACX_OK) {
            // what the ranges did not cover: the tail, and -- when the repair kernels had to place anything the packer had
            // counted as done (never seen) -- everything
            if (e->h_err[kMaxChunks] & 4u) drained_to = 0;
            if (off > drained_to) {
                const hipError_t ce = hipMemcpyAsync(e->h_payload + drained_to, e->d_payload + drained_to, off - drained_to,
                                                     hipMemcpyDeviceToHost, e->copy_stream);
                if (ce != hipSuccess) status = hip_fail(e, ce, "D2H payload");
            }
        }
        hipError_t se = hipStreamSynchronize(e->copy_stream);
        const hipError_t se2 = hipStreamSynchronize(e->copy_stream2);
        if (se == hipSuccess) se = se2;
        if (se != hipSuccess && status == LACX_OK) status = hip_fail(e, se, "D2H payload");
    }
    if (status == -1 && !staged) {
        // no sample-range error can hide behind the overflow: wait for every chunk's block plans first
        (void)hipDeviceSynchronize();
        bool bad = false;
        for (uint32_t b = 0; b < nb; ++b) bad = bad || e->h_bplans[b].invalid;
        if (bad) {
            status = LACX_E_INVALID;
        } else {
            status = reemit_into_regrown_buffer(e, &off);
            e->timing.regrows += 1;
        }
    }
    e->timing.d2h_ms = ms_since(t0);
    if (status != LACX_OK) {
        (void)hipDeviceSynchronize();
        if (status == LACX_E_INVALID) {
            const int rr = check_sample_range(e, nb);
            return rr ? rr : fail(e, LACX_E_INVALID, "sample outside the configured PCM bit depth");
        }
        return status;
    }
    for (size_t c = 0; c < chunks.size(); ++c) {
        add_chunk_timing(e, (int)c);
        float f = 0;
        if (hipEventElapsedTime(&f, e->ev[c][4], e->ev[c][5]) == hipSuccess) e->timing.emit_ms += f;
        (void)hipGetLastError();
    }
    e->timing.full_launches = (uint32_t)chunks.size();
    e->timing.full_slots = (uint64_t)nb * (channels == 2 ? 2u : 1u);
    e->timing.emit_direct = e->timing.moved_by_k_pack = e->timing.packer_gave_up = 0;
    if (e->pend.fused) {
        for (size_t i = 0; i < (size_t)nb * (size_t)channels; ++i) e->timing.emit_direct += e->h_emitted[i] == 1u;
        e->timing.packer_gave_up = e->h_err[kMaxChunks + 2];
        e->timing.moved_by_k_pack = e->h_err[kMaxChunks + 3];
    }
    e->timing.full_exec_ms = 0;
    for (size_t c = 0; c < chunks.size(); ++c) {
        const unsigned long long a = ~e->h_tspan[c], b = e->h_tspan[kMaxChunks + c];  // the start stamp is kept inverted
        if (b > a) e->timing.full_exec_ms += (double)(b - a) * 1e-5;  // 100 MHz device clock -> ms
    }
    *payload_size = off;
    return LACX_OK;
}

int encode_pipelined_device(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, uint64_t frames,
                            hipStream_t user_stream, uint64_t* payload_size, int layout = 0, int layout_channels = 0,
                            const HostSrc* hs = nullptr) {
    const int rc = encode_device_begin(e, d_left, d_right, frames, user_stream, layout, layout_channels, hs);
    if (rc) return rc;
    return encode_device_end(e, payload_size);
}

int fetch_pcm_if_needed(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, uint64_t frames,
                        const int32_t*& h_left, const int32_t*& h_right, std::vector<int32_t>& tl,
                        std::vector<int32_t>& tr) {
    if (h_left) return LACX_OK;
    tl.resize(frames);
    HIP_TRY(e, hipMemcpy(tl.data(), d_left, frames * sizeof(int32_t), hipMemcpyDeviceToHost), "D2H pcm");
    h_left = tl.data();
    if (d_right) {
        tr.resize(frames);
        HIP_TRY(e, hipMemcpy(tr.data(), d_right, frames * sizeof(int32_t), hipMemcpyDeviceToHost), "D2H pcm");
        h_right = tr.data();
    }
    return LACX_OK;
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

// ---- many streams as ONE launch set (lacx_encode_batch_device) ------------------------------------------------------
// Every stream keeps its own parameters (rate, depth, channels, stereo mode, layout); the kernels resolve the stream of a
// block from the descriptor table (StreamDesc, lacx_types.h).  One ingest / Levinson / probe / whole-block launch over
// all blocks of all streams, one packer; every stream's payload lands in its own region of the pinned result buffer.
int encode_batch(lacx_encoder* e, const lacx_batch_item* items, uint32_t n, hipStream_t user_stream, lacx_batch_out* out) {
    if (e->pend.active) return fail(e, LACX_E_RUNTIME, "an encode is already in flight on this encoder");
    std::vector<StreamDesc>& sds = e->batch_streams;
    sds.assign(n, StreamDesc{});
    uint32_t nb = 0, nitems = 0, nwg = 0;
    uint64_t region = 0;
    int max_depth = 16;
    for (uint32_t i = 0; i < n; ++i) {
        const lacx_batch_item& it = items[i];
        const std::string who = "stream " + std::to_string(i) + ": ";
        if (it.pcm.data0 == nullptr || it.frames == 0) return fail(e, LACX_E_INVALID, who + "left channel must not be empty");
        if (!rate_ok(it.sample_rate)) return fail(e, LACX_E_INVALID, who + "unsupported sample rate: " + std::to_string(it.sample_rate));
        if (!(it.bit_depth == 16 || it.bit_depth == 24)) return fail(e, LACX_E_INVALID, who + "unsupported bit depth: " + std::to_string((int)it.bit_depth));
        if (it.stereo_mode > 2) return fail(e, LACX_E_INVALID, who + "unsupported stereo mode: " + std::to_string((int)it.stereo_mode));
        if (it.pcm.channels != 1 && it.pcm.channels != 2) return fail(e, LACX_E_INVALID, who + "unsupported channel count");
        int layout = 0;
        if (it.pcm.layout == LACX_PCM_PLANAR_I32) {
            if ((it.pcm.channels == 2) != (it.pcm.data1 != nullptr))
                return fail(e, LACX_E_INVALID, who + "planar PCM: data1 must be the right channel of stereo input and null for mono");
        } else if (it.pcm.layout == LACX_PCM_INTERLEAVED_I16 || it.pcm.layout == LACX_PCM_INTERLEAVED_I24) {
            if ((it.pcm.layout == LACX_PCM_INTERLEAVED_I16 ? 16 : 24) != it.bit_depth)
                return fail(e, LACX_E_INVALID, who + "PCM layout does not match the bit depth");
            layout = (int)it.pcm.layout;
        } else {
            return fail(e, LACX_E_INVALID, who + "unknown PCM layout");
        }
        const int channels = (int)it.pcm.channels;
        StreamDesc& sd = sds[i];
        sd.prm = make_params(e, it.frames, channels, it.stereo_mode, it.bit_depth, layout);
        sd.prm.stream_base = nitems;
        sd.left = static_cast<const int32_t*>(it.pcm.data0);
        sd.right = layout ? nullptr : static_cast<const int32_t*>(it.pcm.data1);
        sd.first_block = nb;
        sd.first_wg = nwg;
        sd.pad = i;  // the stream's number in the table (k_offsets)
        const uint32_t snb = sd.prm.num_blocks;
        const uint64_t last_frames = it.frames - (uint64_t)(snb - 1) * kMaxBlock;
        const bool last_both_ways = channels == 2 && it.stereo_mode == 2 && last_frames <= (uint64_t)kFullCompareLimit;
        sd.fuse_items = (snb - (last_both_ways ? 1u : 0u)) * (uint32_t)channels;
        sd.out_base = region;
        sd.out_cap = it.frames * (uint64_t)channels * (it.bit_depth / 8u) * 5u / 4u + (uint64_t)snb * 64u + 4096u;
        region += (sd.out_cap + 4095u) & ~4095ull;
        if ((uint64_t)nb + snb > 0x7FFFFFFFull / kSlotsPerBlock) return fail(e, LACX_E_INVALID, "too many blocks in one batch");
        nb += snb;
        nitems += snb * (uint32_t)channels;
        nwg += snb * (uint32_t)channels;
        max_depth = std::max(max_depth, (int)it.bit_depth);
    }
    if (n > 65535u) return fail(e, LACX_E_INVALID, "more than 65535 streams in one batch");
    int rc = ensure_workspace(e, nb);
    if (rc) return rc;
    // staging slots: one stride for the whole set (the deepest material's)
    {
        const int save = e->cfg.bit_depth;
        (void)save;
        rc = ensure_slots(e, nitems, 1, max_depth);
        if (rc) return rc;
    }
    if (region > e->h_payload_cap) {
        if (e->h_payload_base) (void)hipHostFree(e->h_payload_base);
        e->h_payload = e->h_payload_base = nullptr;
        e->h_payload_cap = e->h_prefix = 0;
        HIP_TRY(e, hipHostMalloc((void**)&e->h_payload_base, region, 0), "hipHostMalloc(payload)");
        e->h_payload = e->h_payload_base;
        e->h_payload_cap = region;
    }
    if (nb > e->h_table_blocks) {
        if (e->h_table) (void)hipHostFree(e->h_table);
        e->h_table = nullptr;
        HIP_TRY(e, hipHostMalloc((void**)&e->h_table, (size_t)nb * 2 * sizeof(uint32_t), 0), "hipHostMalloc(table)");
        e->h_table_blocks = nb;
    }
    if (nb * 2u > e->h_emitted_cap) {
        if (e->h_emitted) (void)hipHostFree(e->h_emitted);
        e->h_emitted = nullptr;
        e->h_emitted_cap = 0;
        HIP_TRY(e, hipHostMalloc((void**)&e->h_emitted, (size_t)nb * 2 * sizeof(uint32_t), 0), "hipHostMalloc(emitted)");
        e->h_emitted_cap = nb * 2u;
    }
    // descriptor table + the stream of every stream index -> device
    const size_t tab_bytes = ((size_t)n * sizeof(StreamDesc) + 15) & ~(size_t)15, map_bytes = (size_t)nitems * sizeof(uint16_t);
    if (tab_bytes + map_bytes > e->d_batch_cap) {
        if (e->d_batch) (void)hipFree(e->d_batch);
        if (e->d_wide) (void)hipFree(e->d_wide);
        if (e->h_range) (void)hipHostFree(e->h_range);
        if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
        if (e->copy_stream2) (void)hipStreamDestroy(e->copy_stream2);
        e->d_batch = nullptr;
        e->d_batch_cap = 0;
        HIP_TRY(e, hipMalloc((void**)&e->d_batch, tab_bytes + map_bytes), "hipMalloc(batch table)");
        e->d_batch_cap = tab_bytes + map_bytes;
    }
    std::vector<uint16_t> item_stream(nitems);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t cnt = sds[i].prm.num_blocks * (uint32_t)sds[i].prm.channels;
        std::fill(item_stream.begin() + sds[i].prm.stream_base, item_stream.begin() + sds[i].prm.stream_base + cnt, (uint16_t)i);
    }
    reset_device_timing(e);
    e->timing.emit_ms = 0;
    const auto t0 = clk::now();
    hipStream_t s = user_stream ? user_stream : e->stream[0];
    HIP_TRY(e, hipMemcpyAsync(e->d_batch, sds.data(), (size_t)n * sizeof(StreamDesc), hipMemcpyHostToDevice, s), "H2D batch table");
    HIP_TRY(e, hipMemcpyAsync(e->d_batch + tab_bytes, item_stream.data(), map_bytes, hipMemcpyHostToDevice, s), "H2D batch map");
    HIP_TRY(e, hipStreamSynchronize(s), "synchronize");  // (item_stream is a local; the copies are tiny)
    LaunchSet ls;
    ls.br.table = reinterpret_cast<const StreamDesc*>(e->d_batch);
    ls.br.nstreams = n;
    ls.br.total_blocks = nb;
    ls.br.single = StreamDesc{};
    ls.streams = sds.data();
    ls.nstreams = n;
    ls.total_items = nitems;
    ls.item_stream = reinterpret_cast<const uint16_t*>(e->d_batch + tab_bytes);
    uint8_t* emit_dst = nullptr;
    HIP_TRY(e, hipHostGetDevicePointer((void**)&emit_dst, e->h_payload, 0), "hipHostGetDevicePointer");
    HIP_TRY(e, hipMemsetAsync(e->zero_region, 0, e->zero_bytes, s), "memset");
    DeviceWorkspace w = e->ws;
    w.t_first = e->d_tspan;
    w.t_last = e->d_tspan + kMaxChunks;
    FuseArgs fa;
    fa.slots = e->ws.slots;
    fa.slot_stride = e->ws.slot_stride;
    fa.emitted = e->ws.emitted;
    fa.err_flag = w.err_flag;
    fa.size_rec = e->ws.size_rec;
    fa.ready_rec = e->ws.ready_rec;
    auto run = [&]() -> int {
        HIP_TRY(e, launch_analysis(ls, w, s, e->ev[0], &fa, nullptr), "kernel launch");
        const bool packer = !std::getenv("LACX_NO_PACKER");
        if (packer) {
            HIP_TRY(e, hipStreamWaitEvent(e->pack_stream, e->ev[0][2], 0), "stream wait");
            HIP_TRY(e, launch_stream_out(ls, e->ws, emit_dst, e->ws.err_flag + kMaxChunks, e->pack_stream), "packer launch");
            HIP_TRY(e, hipEventRecord(e->pack_done, e->pack_stream), "event record");
        }
        HIP_TRY(e, launch_emit(ls, w, emit_dst, nullptr, nullptr, nullptr, s, true, packer ? e->ws.err_flag + kMaxChunks + 1 : nullptr,
                               nitems, packer ? e->pack_done : nullptr, e->ws.err_flag + kMaxChunks + 3), "emit launch");
        HIP_TRY(e, hipEventRecord(e->ev[0][5], s), "event record");
        GatherList g;
        auto mapped = [](auto* host) -> decltype(host) {
            void* d = nullptr;
            return hipHostGetDevicePointer(&d, host, 0) == hipSuccess ? static_cast<decltype(host)>(d) : nullptr;
        };
        BlockPlan* m_bplans = mapped(e->h_bplans);
        uint32_t *m_table = mapped(e->h_table), *m_err = mapped(e->h_err), *m_emitted = mapped(e->h_emitted);
        unsigned long long* m_tspan = mapped(e->h_tspan);
        if (!m_bplans || !m_table || !m_err || !m_tspan || !m_emitted) return fail(e, LACX_E_RUNTIME, "hipHostGetDevicePointer failed");
        g.add(w.bplans, m_bplans, (size_t)nb * sizeof(BlockPlan));
        g.add(w.table, m_table, (size_t)nb * 2 * sizeof(uint32_t));
        g.add(w.err_flag, &m_err[0], sizeof(uint32_t));
        g.add(e->ws.err_flag + kMaxChunks, &m_err[kMaxChunks], 4 * sizeof(uint32_t));
        g.add(e->ws.packed, m_emitted, (size_t)nitems * sizeof(uint32_t));
        g.add(w.t_first, &m_tspan[0], sizeof(unsigned long long));
        g.add(w.t_last, &m_tspan[kMaxChunks], sizeof(unsigned long long));
        HIP_TRY(e, launch_gather(g, s), "gather launch");
        HIP_TRY(e, hipEventRecord(e->done[0], s), "event record");
        HIP_TRY(e, hipEventSynchronize(e->done[0]), "event synchronize");
        return LACX_OK;
    };
    rc = run();
    if (rc != LACX_OK) {
        (void)hipDeviceSynchronize();
        return rc;
    }
    e->timing.d2h_ms = ms_since(t0);
    for (uint32_t i = 0; i < n; ++i) {  // sample-range errors, stream by stream, the reference's wording per stream
        const StreamDesc& sd = sds[i];
        for (int pass = 0; pass < 2; ++pass) {
            for (uint32_t b = 0; b < sd.prm.num_blocks; ++b) {
                const BlockPlan& bp = e->h_bplans[sd.first_block + b];
                if (!bp.invalid) continue;
                const bool is_right = (bp.first_bad >> 31) != 0;
                if ((pass == 0) == is_right) continue;
                const uint64_t idx = (uint64_t)b * kMaxBlock + (bp.first_bad & 0x7FFFFFFFu);
                return fail(e, LACX_E_INVALID, "stream " + std::to_string(i) + ": " + (is_right ? "right" : "left") +
                                                   " sample at index " + std::to_string(idx) + " is outside the configured PCM bit depth");
            }
        }
    }
    if (e->h_err[0] & 1u) return fail(e, LACX_E_RUNTIME, "device emit disagrees with the analysis plan (internal error)");
    if ((e->h_err[0] & 2u) || (e->h_err[kMaxChunks] & 2u))
        return fail(e, LACX_E_RUNTIME, "a stream's payload exceeds its pinned result reservation");
    add_chunk_timing(e, 0);
    {
        float f = 0;
        if (hipEventElapsedTime(&f, e->ev[0][4], e->ev[0][5]) == hipSuccess) e->timing.emit_ms += f;
        (void)hipGetLastError();
    }
    e->timing.full_launches = 1;
    e->timing.full_slots = nitems;
    e->timing.emit_direct = 0;
    for (uint32_t i = 0; i < nitems; ++i) e->timing.emit_direct += e->h_emitted[i] == 1u;
    e->timing.packer_gave_up = e->h_err[kMaxChunks + 2];
    e->timing.moved_by_k_pack = e->h_err[kMaxChunks + 3];
    {
        const unsigned long long a = ~e->h_tspan[0], b = e->h_tspan[kMaxChunks];
        e->timing.full_exec_ms = b > a ? (double)(b - a) * 1e-5 : 0.0;
    }
    for (uint32_t i = 0; i < n; ++i) {
        const StreamDesc& sd = sds[i];
        uint64_t bytes = 0;
        for (uint32_t b = 0; b < sd.prm.num_blocks; ++b) {
            const uint32_t by = e->h_table[2 * ((size_t)sd.first_block + b) + 1];
            if (by == 0) return fail(e, LACX_E_RUNTIME, "encoded block size is outside format limits");
            bytes += by;
        }
        if (bytes > sd.out_cap) return fail(e, LACX_E_RUNTIME, "a stream's payload exceeds its pinned result reservation");
        out[i].payload = e->h_payload + sd.out_base;
        out[i].payload_size = bytes;
        out[i].table = e->h_table + 2 * (size_t)sd.first_block;
        out[i].nblocks = sd.prm.num_blocks;
        out[i].reserved = 0;
    }
    e->timing.total_ms = ms_since(t0);
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

}  // namespace

extern "C" {

int lacx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lacx_encoder_create(const lacx_config* cfg, lacx_encoder** out) {
    if (!cfg || !out) return LACX_E_INVALID;
    lacx_encoder* e = new lacx_encoder();
    e->cfg = *cfg;
    *out = e;
    return LACX_OK;
}

void lacx_encoder_destroy(lacx_encoder* e) {
    if (!e) return;
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
        if (e->h_tspan) (void)hipHostFree(e->h_tspan);

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

int lacx_analyze_device(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, uint64_t frames,
                        void* stream, lacx_block_plan* bplans, lacx_channel_plan* plans) {
    if (!e) return LACX_E_INVALID;
    int rc = prepare(e, d_left, frames);
    if (rc) return rc;
    const int channels = d_right ? 2 : 1;
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : e->stream[0];
    rc = analyze_on_device(e, d_left, d_right, frames, channels, e->cfg.stereo_mode, e->cfg.bit_depth, st);
    if (rc) return rc;
    const uint32_t nb = blocks_for(frames);
    if (bplans) std::memcpy(bplans, e->h_bplans, (size_t)nb * sizeof(BlockPlan));
    if (plans) std::memcpy(plans, e->h_plans, (size_t)nb * kSlotsPerBlock * sizeof(ChannelPlan));
    return check_sample_range(e, nb);
}

int lacx_analyze(lacx_encoder* e, const int32_t* left, const int32_t* right, uint64_t frames,
                 lacx_block_plan* bplans, lacx_channel_plan* plans) {
    if (!e) return LACX_E_INVALID;
    int rc = prepare(e, left, frames);
    if (rc) return rc;
    rc = upload(e, left, right, frames);
    if (rc) return rc;
    return lacx_analyze_device(e, e->d_left, right ? e->d_right : nullptr, frames, nullptr, bplans, plans);
}

int lacx_emit_from_plans(lacx_encoder* e, const int32_t* left, const int32_t* right, uint64_t frames,
                         const lacx_block_plan* bplans_c, const lacx_channel_plan* plans_c, uint8_t** out,
                         uint64_t* out_size) {
    if (!e || !out || !out_size || !bplans_c || !plans_c) return LACX_E_INVALID;
    const int rc = validate_stream_args(e, left, frames);
    if (rc) return rc;
    const BlockPlan* bplans = reinterpret_cast<const BlockPlan*>(bplans_c);
    const ChannelPlan* plans = reinterpret_cast<const ChannelPlan*>(plans_c);
    const int channels = right ? 2 : 1;
    const uint32_t nb = blocks_for(frames);
    const StreamParams sp = stream_params(e->cfg, channels);
    std::vector<uint64_t> offsets((size_t)nb + 1, 0);
    uint64_t off = 0;
    for (uint32_t b = 0; b < nb; ++b) {
        offsets[b] = off;
        off += block_payload_bytes(sp, bplans[b], plans + (size_t)b * kSlotsPerBlock);
    }
    offsets[nb] = off;
    const uint64_t head = 10 + 4 + 8ull * nb;
    uint8_t* buf = static_cast<uint8_t*>(std::malloc(head + off));
    if (!buf) return fail(e, LACX_E_RUNTIME, "out of memory");
    write_frame_header(sp, buf);
    put32(buf + 10, nb);
    for (uint32_t b = 0; b < nb; ++b) {
        const uint64_t size = offsets[b + 1] - offsets[b];
        if (size == 0 || size > 0xFFFFFFFFull) {
            std::free(buf);
            return fail(e, LACX_E_RUNTIME, "encoded block size is outside format limits");
        }
        put32(buf + 14 + 8ull * b, bplans[b].frames);
        put32(buf + 18 + 8ull * b, (uint32_t)size);
    }
    const std::string err = emit_blocks(sp, left, right, frames, bplans, plans, nb, offsets.data(), buf + head, off,
                                        e->cfg.emit_threads);
    if (!err.empty()) {
        std::free(buf);
        return fail(e, LACX_E_RUNTIME, err);
    }
    *out = buf;
    *out_size = head + off;
    return LACX_OK;
}

int lacx_encode_device(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, const int32_t* h_left,
                       const int32_t* h_right, uint64_t frames, void* stream, uint8_t** out,
                       uint64_t* out_size) {
    if (!e || !out || !out_size) return LACX_E_INVALID;
    const auto t0 = clk::now();
    const double h2d = e->timing.h2d_ms;
    e->timing = lacx_timing{};
    e->timing.h2d_ms = h2d;
    int rc = prepare(e, d_left, frames);
    if (rc) return rc;
    const uint32_t nb = blocks_for(frames);
    const uint64_t head = 10 + 4 + 8ull * nb;
    if (!(e->cfg.flags & LACX_FLAG_HOST_EMIT)) {
        uint64_t pay = 0;
        rc = encode_pipelined_device(e, d_left, d_right, frames, static_cast<hipStream_t>(stream), &pay);
        if (rc == LACX_OK) {
            uint8_t* buf = static_cast<uint8_t*>(std::malloc(head + pay));
            if (!buf) return fail(e, LACX_E_RUNTIME, "out of memory");
            write_frame_header(stream_params(e->cfg, d_right ? 2 : 1), buf);
            put32(buf + 10, nb);
            for (uint32_t b = 0; b < nb; ++b) {
                if (e->h_table[2 * b + 1] == 0) {
                    std::free(buf);
                    return fail(e, LACX_E_RUNTIME, "encoded block size is outside format limits");
                }
                put32(buf + 14 + 8ull * b, e->h_table[2 * b]);
                put32(buf + 18 + 8ull * b, e->h_table[2 * b + 1]);
            }
            big_copy(buf + head, e->h_payload, pay);
            *out = buf;
            *out_size = head + pay;
            e->timing.total_ms = ms_since(t0);
            return LACX_OK;
        }
        if (rc != -1) return rc;  // -1: pinned reservation too small -> host-emit pipeline below
    }
    std::vector<int32_t> tl, tr;
    rc = fetch_pcm_if_needed(e, d_left, d_right, frames, h_left, h_right, tl, tr);
    if (rc) return rc;
    uint8_t* buf = nullptr;
    uint64_t pay = 0;
    std::vector<uint64_t> offsets;
    rc = encode_pipelined(e, d_left, d_right, h_left, d_right ? h_right : nullptr, frames,
                          static_cast<hipStream_t>(stream), head, &buf, &pay, offsets);
    if (rc) return rc;
    write_frame_header(stream_params(e->cfg, d_right ? 2 : 1), buf);
    rc = fill_table(e, buf, nb, offsets);
    if (rc) {
        std::free(buf);
        return rc;
    }
    *out = buf;
    *out_size = head + pay;
    e->timing.total_ms = ms_since(t0);
    e->timing.emit_ms = e->timing.total_ms - e->timing.d2h_ms;
    return LACX_OK;
}

int lacx_encode(lacx_encoder* e, const int32_t* left, const int32_t* right, uint64_t frames, uint8_t** out,
                uint64_t* out_size) {
    if (!e || !out || !out_size) return LACX_E_INVALID;
    const auto t0 = clk::now();
    e->timing = lacx_timing{};
    int rc = prepare(e, left, frames);
    if (rc) return rc;
    if (!(e->cfg.flags & LACX_FLAG_HOST_EMIT)) {
        // device emit: the upload is pipelined with the analysis (chunk c+1's PCM crosses PCIe under chunk c's kernels)
        rc = ensure_pcm(e, frames, right != nullptr);
        if (rc) return rc;
        HostSrc hs;
        hs.p0 = reinterpret_cast<const uint8_t*>(left);
        hs.p1 = reinterpret_cast<const uint8_t*>(right);
        hs.frame_bytes = sizeof(int32_t);
        uint64_t pay = 0;
        rc = encode_pipelined_device(e, e->d_left, right ? e->d_right : nullptr, frames, nullptr, &pay, 0, 0, &hs);
        if (rc == LACX_OK) {
            const uint32_t nb = blocks_for(frames);
            const uint64_t head = 10 + 4 + 8ull * nb;
            uint8_t* lac = e->h_payload - head;  // h_prefix >= head bytes are reserved in front of the payload
            write_frame_header(stream_params(e->cfg, right ? 2 : 1), lac);
            put32(lac + 10, nb);
            for (uint32_t b = 0; b < nb; ++b) {
                if (e->h_table[2 * b + 1] == 0) return fail(e, LACX_E_RUNTIME, "encoded block size is outside format limits");
                put32(lac + 14 + 8ull * b, e->h_table[2 * b]);
                put32(lac + 18 + 8ull * b, e->h_table[2 * b + 1]);
            }
            uint8_t* buf = static_cast<uint8_t*>(std::malloc(head + pay));
            if (!buf) return fail(e, LACX_E_RUNTIME, "out of memory");
            big_copy(buf, lac, head + pay);
            *out = buf;
            *out_size = head + pay;
            e->timing.total_ms = ms_since(t0);
            return LACX_OK;
        }
        if (rc != -1) return rc;
        // -1 (only with LACX_EMIT_STAGED): fall through to the host-emit pipeline; the PCM is on the device already
        rc = lacx_encode_device(e, e->d_left, right ? e->d_right : nullptr, left, right, frames, nullptr, out, out_size);
        e->timing.total_ms = ms_since(t0);
        return rc;
    }
    rc = upload(e, left, right, frames);
    if (rc) return rc;
    rc = lacx_encode_device(e, e->d_left, right ? e->d_right : nullptr, left, right, frames, nullptr, out, out_size);
    e->timing.total_ms = ms_since(t0);
    return rc;
}

static int shard_host_path(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, const int32_t* h_left,
                           const int32_t* h_right, uint64_t frames, void* stream, uint8_t** payload,
                           uint64_t* payload_size, uint32_t** table, uint32_t* nblocks) {
    std::vector<int32_t> tl, tr;
    int rc = fetch_pcm_if_needed(e, d_left, d_right, frames, h_left, h_right, tl, tr);
    if (rc) return rc;
    const uint32_t nb = blocks_for(frames);
    uint8_t* buf = nullptr;
    uint64_t pay = 0;
    std::vector<uint64_t> offsets;
    rc = encode_pipelined(e, d_left, d_right, h_left, d_right ? h_right : nullptr, frames,
                          static_cast<hipStream_t>(stream), 0, &buf, &pay, offsets);
    if (rc) return rc;
    uint32_t* tab = static_cast<uint32_t*>(std::malloc(sizeof(uint32_t) * 2 * (nb ? nb : 1)));
    if (!tab) {
        std::free(buf);
        return fail(e, LACX_E_RUNTIME, "out of memory");
    }
    for (uint32_t b = 0; b < nb; ++b) {
        tab[2 * b] = e->h_bplans[b].frames;
        tab[2 * b + 1] = (uint32_t)(offsets[b + 1] - offsets[b]);
    }
    *payload = buf;
    *payload_size = pay;
    *table = tab;
    *nblocks = nb;
    return LACX_OK;
}

int lacx_encode_shard_device_view(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right,
                                  const int32_t* h_left, const int32_t* h_right, uint64_t frames, void* stream,
                                  const uint8_t** payload, uint64_t* payload_size, const uint32_t** table,
                                  uint32_t* nblocks) {
    if (!e || !payload || !payload_size || !table || !nblocks) return LACX_E_INVALID;
    const auto t0 = clk::now();
    const double h2d = e->timing.h2d_ms;
    e->timing = lacx_timing{};
    e->timing.h2d_ms = h2d;
    int rc = prepare(e, d_left, frames);
    if (rc) return rc;
    if (!(e->cfg.flags & LACX_FLAG_HOST_EMIT)) {
        uint64_t pay = 0;
        rc = encode_pipelined_device(e, d_left, d_right, frames, static_cast<hipStream_t>(stream), &pay);
        if (rc == LACX_OK) {
            *payload = e->h_payload;
            *payload_size = pay;
            *table = e->h_table;
            *nblocks = blocks_for(frames);
            e->timing.total_ms = ms_since(t0);
            return LACX_OK;
        }
        if (rc != -1) return rc;
    }
    std::free(e->view_buf);
    std::free(e->view_table);
    e->view_buf = nullptr;
    e->view_table = nullptr;
    uint8_t* buf = nullptr;
    uint32_t* tab = nullptr;
    rc = shard_host_path(e, d_left, d_right, h_left, h_right, frames, stream, &buf, payload_size, &tab, nblocks);
    if (rc) return rc;
    e->view_buf = buf;
    e->view_table = tab;
    *payload = buf;
    *table = tab;
    e->timing.total_ms = ms_since(t0);
    e->timing.emit_ms = e->timing.total_ms - e->timing.d2h_ms;
    return LACX_OK;
}

int lacx_encode_shard_pcm_device_begin(lacx_encoder* e, const lacx_pcm* pcm, uint64_t frames, void* stream) {
    if (!e || !pcm) return LACX_E_INVALID;
    e->timing = lacx_timing{};
    int rc = prepare(e, pcm->data0, frames);
    if (rc) return rc;
    if (pcm->channels != 1 && pcm->channels != 2) return fail(e, LACX_E_INVALID, "unsupported channel count");
    if (e->cfg.flags & LACX_FLAG_HOST_EMIT)
        return fail(e, LACX_E_INVALID, "this entry point needs the device-side emit (LACX_FLAG_HOST_EMIT is set)");
    if (pcm->layout == LACX_PCM_PLANAR_I32) {
        if ((pcm->channels == 2) != (pcm->data1 != nullptr))
            return fail(e, LACX_E_INVALID, "planar PCM: data1 must be the right channel of stereo input and null for mono");
        return encode_device_begin(e, static_cast<const int32_t*>(pcm->data0), static_cast<const int32_t*>(pcm->data1), frames,
                                   static_cast<hipStream_t>(stream));
    }
    const int want_depth = pcm->layout == LACX_PCM_INTERLEAVED_I16 ? 16 : (pcm->layout == LACX_PCM_INTERLEAVED_I24 ? 24 : 0);
    if (want_depth == 0) return fail(e, LACX_E_INVALID, "unknown PCM layout");
    if (e->cfg.bit_depth != want_depth) return fail(e, LACX_E_INVALID, "PCM layout does not match the configured bit depth");
    return encode_device_begin(e, static_cast<const int32_t*>(pcm->data0), nullptr, frames, static_cast<hipStream_t>(stream),
                               (int)pcm->layout, (int)pcm->channels);
}

int lacx_encode_shard_end(lacx_encoder* e, const uint8_t** payload, uint64_t* payload_size, const uint32_t** table,
                          uint32_t* nblocks) {
    if (!e || !payload || !payload_size || !table || !nblocks) return LACX_E_INVALID;
    const uint32_t nb = e->pend.nb;
    uint64_t pay = 0;
    const int rc = encode_device_end(e, &pay);
    if (rc == -1) return fail(e, LACX_E_RUNTIME, "payload exceeds the pinned result reservation");
    if (rc) return rc;
    *payload = e->h_payload;
    *payload_size = pay;
    *table = e->h_table;
    *nblocks = nb;
    e->timing.total_ms = ms_since(e->pend.t0);
    return LACX_OK;
}

int lacx_encode_shard_pcm_device_view(lacx_encoder* e, const lacx_pcm* pcm, uint64_t frames, void* stream,
                                      const uint8_t** payload, uint64_t* payload_size, const uint32_t** table,
                                      uint32_t* nblocks) {
    if (!e || !pcm || !payload || !payload_size || !table || !nblocks) return LACX_E_INVALID;
    if (pcm->layout == LACX_PCM_PLANAR_I32)
        return lacx_encode_shard_device_view(e, static_cast<const int32_t*>(pcm->data0),
                                             static_cast<const int32_t*>(pcm->data1), nullptr, nullptr, frames, stream,
                                             payload, payload_size, table, nblocks);
    const int rc = lacx_encode_shard_pcm_device_begin(e, pcm, frames, stream);
    if (rc) return rc;
    return lacx_encode_shard_end(e, payload, payload_size, table, nblocks);
}

int lacx_encode_batch_device(lacx_encoder* e, const lacx_batch_item* items, uint32_t n, void* stream, lacx_batch_out* out) {
    if (!e || !items || !out || n == 0) return LACX_E_INVALID;
    e->timing = lacx_timing{};
    if (e->cfg.flags & LACX_FLAG_HOST_EMIT)
        return fail(e, LACX_E_INVALID, "this entry point needs the device-side emit (LACX_FLAG_HOST_EMIT is set)");
    int rc = ensure_device(e);
    if (rc) return rc;
    HIP_TRY(e, hipSetDevice(e->device), "hipSetDevice");
    return encode_batch(e, items, n, static_cast<hipStream_t>(stream), out);
}

int lacx_encode_shard_device(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right,
                             const int32_t* h_left, const int32_t* h_right, uint64_t frames, void* stream,
                             uint8_t** payload, uint64_t* payload_size, uint32_t** table, uint32_t* nblocks) {
    if (!e || !payload || !payload_size || !table || !nblocks) return LACX_E_INVALID;
    const uint8_t* vp = nullptr;
    const uint32_t* vt = nullptr;
    uint64_t pay = 0;
    uint32_t nb = 0;
    const int rc = lacx_encode_shard_device_view(e, d_left, d_right, h_left, h_right, frames, stream, &vp, &pay, &vt, &nb);
    if (rc) return rc;
    uint8_t* buf = static_cast<uint8_t*>(std::malloc(pay ? pay : 1));
    uint32_t* tab = static_cast<uint32_t*>(std::malloc(sizeof(uint32_t) * 2 * (nb ? nb : 1)));
    if (!buf || !tab) {
        std::free(buf);
        std::free(tab);
        return fail(e, LACX_E_RUNTIME, "out of memory");
    }
    big_copy(buf, vp, pay);
    std::memcpy(tab, vt, sizeof(uint32_t) * 2 * nb);
    *payload = buf;
    *payload_size = pay;
    *table = tab;
    *nblocks = nb;
    return LACX_OK;
}

int lacx_encode_shard(lacx_encoder* e, const int32_t* left, const int32_t* right, uint64_t frames,
                      uint8_t** payload, uint64_t* payload_size, uint32_t** table, uint32_t* nblocks) {
    if (!e) return LACX_E_INVALID;
    e->timing = lacx_timing{};
    int rc = prepare(e, left, frames);
    if (rc) return rc;
    rc = upload(e, left, right, frames);
    if (rc) return rc;
    return lacx_encode_shard_device(e, e->d_left, right ? e->d_right : nullptr, left, right, frames, nullptr,
                                    payload, payload_size, table, nblocks);
}

int lacx_wav_parse(const uint8_t* wav, uint64_t size, lacx_wav_info* out) {
    if (!wav || !out) return LACX_E_INVALID;
    WavInfo w;
    if (!wav_parse(wav, size, &w)) return LACX_E_INVALID;
    out->channels = w.channels;
    out->bit_depth = w.bit_depth;
    out->sample_rate = w.sample_rate;
    out->frames = w.frames;
    out->data_offset = w.data_offset;
    out->data_bytes = w.data_bytes;
    return LACX_OK;
}

// WAV image in host memory -> complete .lac in the encoder's pinned result buffer (header and block table are written
// in front of the payload, which the device put there itself): no copy of the result at all.
static int encode_wav_in_place(lacx_encoder* e, const uint8_t* wav, uint64_t size, const uint8_t** out, uint64_t* out_size) {
    const auto t0 = clk::now();
    e->timing = lacx_timing{};
    WavInfo w;
    if (!wav || !wav_parse(wav, size, &w)) return fail(e, LACX_E_INVALID, "not a supported PCM WAV file");
    if (w.sample_rate != e->cfg.sample_rate || w.bit_depth != e->cfg.bit_depth)
        return fail(e, LACX_E_INVALID, "WAV format (" + std::to_string(w.sample_rate) + " Hz, " +
                                           std::to_string((int)w.bit_depth) + " bit) differs from the encoder's");
    int rc = prepare(e, wav + w.data_offset, w.frames);
    if (rc) return rc;
    if (w.data_bytes + 16u > e->d_raw_cap) {
        if (e->d_raw) (void)hipFree(e->d_raw);
        e->d_raw = nullptr;
        e->d_raw_cap = 0;
        HIP_TRY(e, hipMalloc((void**)&e->d_raw, w.data_bytes + 16u), "hipMalloc(wav data)");
        e->d_raw_cap = w.data_bytes + 16u;
    }
    // The data chunk as it is in the file: interleaved little-endian int16 / packed int24 (coalesced ingest), uploaded
    // chunk by chunk in front of each pipeline chunk's kernels (the upload of chunk c+1 overlaps the analysis of chunk c).
    HostSrc hs;
    hs.p0 = wav + w.data_offset;
    hs.frame_bytes = (uint64_t)w.channels * (w.bit_depth / 8u);
    const int layout = w.bit_depth == 16 ? (int)LACX_PCM_INTERLEAVED_I16 : (int)LACX_PCM_INTERLEAVED_I24;
    uint64_t pay = 0;
    rc = encode_pipelined_device(e, reinterpret_cast<const int32_t*>(e->d_raw), nullptr, w.frames, nullptr, &pay, layout,
                                 (int)w.channels, &hs);
    if (rc == -1) return fail(e, LACX_E_RUNTIME, "payload exceeds the pinned result reservation");
    if (rc) return rc;
    const uint32_t nb = blocks_for(w.frames);
    const uint64_t head = 10 + 4 + 8ull * nb;
    uint8_t* lac = e->h_payload - head;  // h_prefix >= head bytes are reserved in front of the payload
    write_frame_header(stream_params(e->cfg, (int)w.channels), lac);
    put32(lac + 10, nb);
    for (uint32_t b = 0; b < nb; ++b) {
        if (e->h_table[2 * b + 1] == 0) return fail(e, LACX_E_RUNTIME, "encoded block size is outside format limits");
        put32(lac + 14 + 8ull * b, e->h_table[2 * b]);
        put32(lac + 18 + 8ull * b, e->h_table[2 * b + 1]);
    }
    *out = lac;
    *out_size = head + pay;
    e->timing.total_ms = ms_since(t0);
    return LACX_OK;
}

int lacx_encode_wav_view(lacx_encoder* e, const uint8_t* wav, uint64_t size, const uint8_t** out, uint64_t* out_size) {
    if (!e || !out || !out_size) return LACX_E_INVALID;
    return encode_wav_in_place(e, wav, size, out, out_size);
}

int lacx_encode_wav(lacx_encoder* e, const uint8_t* wav, uint64_t size, uint8_t** out, uint64_t* out_size) {
    if (!e || !out || !out_size) return LACX_E_INVALID;
    const auto t0 = clk::now();
    const uint8_t* view = nullptr;
    uint64_t n = 0;
    const int rc = encode_wav_in_place(e, wav, size, &view, &n);
    if (rc) return rc;
    uint8_t* buf = static_cast<uint8_t*>(std::malloc(n ? n : 1));
    if (!buf) return fail(e, LACX_E_RUNTIME, "out of memory");
    big_copy(buf, view, n);
    *out = buf;
    *out_size = n;
    e->timing.total_ms = ms_since(t0);
    return LACX_OK;
}

// ---- decode -------------------------------------------------------------------------------------
namespace {
thread_local std::string g_decode_err;
int decode_fail(int code, const std::string& msg) {
    g_decode_err = msg;
    return code;
}
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
}  // namespace

const char* lacx_decode_last_error(void) { return g_decode_err.c_str(); }

// Container header + block table: the structural rules of the reference's reader (src/codec/frame/frame_header.hpp:48-74,
// lac/decoder.cpp:84-145) -- sync, version 3, channels, stereo mode (0 for mono), one of the four sample rates, depth,
// reserved byte; at least one block; every block 1..16384 frames, non-final ones at least 256; non-zero compressed
// sizes that add up to the file; at most 6 912 000 000 samples and a WAV that RIFF can hold.  NOT taken over: its cap on
// the decoded PCM (1 GiB) and the block count that follows from it, which would refuse the 2 h stream of BASELINE
// configs[3].  The legacy version-2 container (no compressed sizes, hence no parallelism) is read too: one lane walks it.
int lacx_stream_parse(const uint8_t* lac, uint64_t size, lacx_stream_info* out) {
    if (!lac || !out) return decode_fail(LACX_E_INVALID, "null argument");
    if (size == 0) return decode_fail(LACX_E_INVALID, "[decode-error] empty input");
    if (size < 10 || lac[0] != 0x4C || lac[1] != 0x41 || (lac[2] != 3 && lac[2] != 2))
        return decode_fail(LACX_E_INVALID, "[decode-error] invalid frame header");
    const int version = lac[2], ch = lac[3], sm = lac[4], bd = lac[8];
    const uint32_t sr = ((uint32_t)lac[5] << 8) | lac[6] | ((uint32_t)lac[7] << 16);
    const bool rate_ok = sr == 44100 || sr == 48000 || sr == 96000 || sr == 192000;
    if ((ch != 1 && ch != 2) || sm > 2 || (ch == 1 && sm != 0) || !rate_ok || (bd != 16 && bd != 24) || lac[9] != 0)
        return decode_fail(LACX_E_INVALID, "[decode-error] invalid frame header");
    if (size < 14) return decode_fail(LACX_E_INVALID, "[decode-error] invalid block count");
    const uint32_t nb = be32(lac + 10);
    if (nb == 0) return decode_fail(LACX_E_INVALID, "[decode-error] invalid block count");
    const uint64_t entry = version >= 3 ? 8u : 4u;  // version 2 has no compressed sizes (ref lac/decoder.cpp:100-104)
    if (size < 14 + entry * nb) return decode_fail(LACX_E_INVALID, "[decode-error] truncated block size table");
    uint64_t frames = 0, pay = 0;
    for (uint32_t b = 0; b < nb; ++b) {
        const uint32_t n = be32(lac + 14 + entry * b);
        if (n == 0 || n > (uint32_t)kMaxBlock || (b + 1 < nb && n < 256u)) return decode_fail(LACX_E_INVALID, "[decode-error] invalid block size");
        frames += n;
        if (frames > 6912000000ull) return decode_fail(LACX_E_INVALID, "[decode-error] total samples exceed maximum");
        if (version >= 3) {
            const uint32_t by = be32(lac + 18 + 8ull * b);
            // The device reader's bit positions are 32-bit and relative to the block: a block must stay below 2^29 bytes.
            // (The reference takes any non-zero size that fits the file; a block this long -- a Rice token at k = 0 may
            // carry a unary part of up to 2^30 bits -- is a documented deviation, see lacx.h.)
            if (by == 0 || by >= (1u << 29)) return decode_fail(LACX_E_INVALID, "[decode-error] invalid compressed block size");
            pay += by;
            if (pay > size) return decode_fail(LACX_E_INVALID, "[decode-error] compressed block sizes exceed frame payload");
        }
    }
    const uint64_t wav_bytes = frames * (uint64_t)ch * (uint64_t)(bd / 8);
    if (36u + wav_bytes + (wav_bytes & 1u) > 0xFFFFFFFFull) return decode_fail(LACX_E_INVALID, "[decode-error] decoded WAV data exceeds RIFF limit");
    if (version >= 3 && 14 + 8ull * nb + pay != size) return decode_fail(LACX_E_INVALID, "[decode-error] block payloads do not fill the file");
    if (version == 2 && size - (14 + 4ull * nb) >= (1ull << 29)) return decode_fail(LACX_E_INVALID, "[decode-error] version-2 payload too large for the serial reader");
    out->sample_rate = sr;
    out->blocks = nb;
    out->frames = frames;
    out->channels = (uint8_t)ch;
    out->bit_depth = (uint8_t)bd;
    out->stereo_mode = (uint8_t)sm;
    out->version = (uint8_t)version;
    return LACX_OK;
}

int lacx_decode(int device, const uint8_t* lac, uint64_t size, int32_t* left, int32_t* right, uint64_t frames,
                float* device_ms) {
    lacx_stream_info info;
    const int prc = lacx_stream_parse(lac, size, &info);
    if (prc) return prc;
    if (!left || (info.channels == 2 && !right)) return decode_fail(LACX_E_INVALID, "output arrays missing");
    if (frames != info.frames) return decode_fail(LACX_E_INVALID, "output arrays do not match the stream's frame count");
    if (device_ms) *device_ms = 0.f;
    if (lacx_device_count() <= 0) return decode_fail(LACX_E_DEVICE, "no usable HIP device");
#define DEC_TRY(call, what)                                                                                  \
    do {                                                                                                     \
        const hipError_t _e = (call);                                                                        \
        if (_e != hipSuccess) {                                                                              \
            rc = decode_fail(LACX_E_DEVICE, std::string(what) + ": " + hipGetErrorString(_e));               \
            goto done;                                                                                       \
        }                                                                                                    \
    } while (0)
    int rc = LACX_OK;
    const uint32_t nb = info.blocks;
    const bool v2 = info.version == 2;
    const uint64_t entry = v2 ? 4u : 8u;
    const uint64_t head = 14 + entry * nb, pay = size - head;
    std::vector<unsigned long long> offs(2 * ((size_t)nb + 1));  // byte offsets, then frame offsets
    unsigned long long* byte_off = offs.data();
    unsigned long long* frame_off = offs.data() + nb + 1;
    byte_off[0] = frame_off[0] = 0;
    for (uint32_t b = 0; b < nb; ++b) {
        frame_off[b + 1] = frame_off[b] + be32(lac + 14 + entry * b);
        byte_off[b + 1] = v2 ? 0 : byte_off[b] + be32(lac + 18 + 8ull * b);
    }
    uint8_t* d_pay = nullptr;
    unsigned long long* d_offs = nullptr;
    int32_t *d_left = nullptr, *d_right = nullptr;
    uint32_t* d_status = nullptr;
    uint8_t* d_ms = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<uint32_t> status(nb);
    int prev_device = -1;  // the caller's current device is put back on the way out
    if (device >= 0) {
        DEC_TRY(hipGetDevice(&prev_device), "hipGetDevice");
        if (prev_device == device) prev_device = -1;
        else DEC_TRY(hipSetDevice(device), "hipSetDevice");
    }
    DEC_TRY(hipMalloc((void**)&d_pay, pay + kDecodeTailPad), "hipMalloc(payload)");  // the bit reader's look-ahead (decode.hip)
    DEC_TRY(hipMemset(d_pay + pay, 0, kDecodeTailPad), "memset");
    DEC_TRY(hipMemcpy(d_pay, lac + head, pay, hipMemcpyHostToDevice), "H2D payload");
    DEC_TRY(hipMalloc((void**)&d_offs, offs.size() * sizeof(unsigned long long)), "hipMalloc(offsets)");
    DEC_TRY(hipMemcpy(d_offs, offs.data(), offs.size() * sizeof(unsigned long long), hipMemcpyHostToDevice), "H2D offsets");
    DEC_TRY(hipMalloc((void**)&d_left, frames * sizeof(int32_t)), "hipMalloc(left)");
    if (info.channels == 2) DEC_TRY(hipMalloc((void**)&d_right, frames * sizeof(int32_t)), "hipMalloc(right)");
    DEC_TRY(hipMalloc((void**)&d_status, (size_t)nb * sizeof(uint32_t)), "hipMalloc(status)");
    DEC_TRY(hipMalloc((void**)&d_ms, nb), "hipMalloc(flags)");
    DEC_TRY(hipEventCreate(&e0), "hipEventCreate");
    DEC_TRY(hipEventCreate(&e1), "hipEventCreate");
    DEC_TRY(hipEventRecord(e0, nullptr), "event record");
    if (v2)
        DEC_TRY(launch_decode_serial(nb, info.channels, info.stereo_mode, info.bit_depth, d_pay, (uint32_t)(8ull * pay), d_offs + nb + 1,
                                     d_left, d_right, d_status, d_ms, nullptr), "decode launch");
    else
        DEC_TRY(launch_decode(nb, info.channels, info.stereo_mode, info.bit_depth, d_pay, d_offs, d_offs + nb + 1, d_left, d_right,
                              d_status, d_ms, nullptr), "decode launch");
    DEC_TRY(hipEventRecord(e1, nullptr), "event record");
    DEC_TRY(hipMemcpy(status.data(), d_status, (size_t)nb * sizeof(uint32_t), hipMemcpyDeviceToHost), "D2H status");
    if (device_ms) (void)hipEventElapsedTime(device_ms, e0, e1);
    for (uint32_t b = 0; b < nb; ++b) {
        if (status[b]) {  // the first failing block, like the reference's message (lac/decoder.cpp:24-32)
            static const char* const kWhat[] = {"", "block header", "channel header", "residual", "padding", "sample overflow",
                                                "trailing bytes", "sample outside the bit depth", "not reached", "residual beyond 2^30"};
            rc = decode_fail(LACX_E_RUNTIME, "[decode-error] block=" + std::to_string(b) + " " +
                                                 (status[b] < 10 ? kWhat[status[b]] : "?"));
            goto done;
        }
    }
    DEC_TRY(hipMemcpy(left, d_left, frames * sizeof(int32_t), hipMemcpyDeviceToHost), "D2H left");
    if (info.channels == 2) DEC_TRY(hipMemcpy(right, d_right, frames * sizeof(int32_t), hipMemcpyDeviceToHost), "D2H right");
done:
#undef DEC_TRY
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d_pay) (void)hipFree(d_pay);
    if (d_offs) (void)hipFree(d_offs);
    if (d_left) (void)hipFree(d_left);
    if (d_right) (void)hipFree(d_right);
    if (d_status) (void)hipFree(d_status);
    if (d_ms) (void)hipFree(d_ms);
    if (prev_device >= 0) (void)hipSetDevice(prev_device);
    return rc;
}

int lacx_assemble(const lacx_config* cfg, int channels, uint32_t nshards, const uint8_t* const* payloads,
                  const uint64_t* payload_sizes, const uint32_t* const* tables, const uint32_t* nblocks,
                  uint8_t** out, uint64_t* out_size) {
    if (!cfg || !out || !out_size || (channels != 1 && channels != 2)) return LACX_E_INVALID;
    uint64_t nb = 0, pay = 0;
    for (uint32_t s = 0; s < nshards; ++s) {
        nb += nblocks[s];
        pay += payload_sizes[s];
    }
    if (nb == 0 || nb > 0xFFFFFFFFull) return LACX_E_INVALID;
    const uint64_t head = 10 + 4 + 8 * nb;
    uint8_t* buf = static_cast<uint8_t*>(std::malloc(head + pay));
    if (!buf) return LACX_E_RUNTIME;
    write_frame_header(stream_params(*cfg, channels), buf);
    put32(buf + 10, (uint32_t)nb);
    uint64_t bi = 0, off = head;
    for (uint32_t s = 0; s < nshards; ++s) {
        for (uint32_t b = 0; b < nblocks[s]; ++b, ++bi) {
            if (tables[s][2 * b + 1] == 0) {
                std::free(buf);
                return LACX_E_RUNTIME;
            }
            put32(buf + 14 + 8 * bi, tables[s][2 * b]);
            put32(buf + 18 + 8 * bi, tables[s][2 * b + 1]);
        }
        big_copy(buf + off, payloads[s], payload_sizes[s]);
        off += payload_sizes[s];
    }
    *out = buf;
    *out_size = head + pay;
    return LACX_OK;
}

// Block::Encoder::encode takes any int32 samples (ref src/codec/block/encoder.cpp:313-316).  Blocks inside the 25-bit
// mid/side domain of validated 16 / 24-bit input go through the streaming kernels; anything wider goes through the wide
// kernel (wide.hip: residuals that leave int32 and the reference's order fallback, 32-bit zigzag values, k up to 31).
// What stays out: blocks of more than 16384 samples -- the container cannot carry them (ref src/codec/lac/decoder.cpp
// refuses a block size above Block::MAX_BLOCK_SIZE) and the kernels' images are sized for that maximum.
static int block_size_check(lacx_encoder* e, uint32_t n) {
    if (n > (uint32_t)kMaxBlock) return fail(e, LACX_E_INVALID, "block larger than 16384 samples (the LAC container cannot carry it)");
    return LACX_OK;
}
static bool block_is_wide(const int32_t* pcm, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i)
        if (pcm[i] > (1 << 24) || pcm[i] < -(1 << 24)) return true;
    return false;
}

static int block_analyze(lacx_encoder* e, const int32_t* pcm, uint32_t n) {
    int rc = ensure_device(e);
    if (rc) return rc;
    HIP_TRY(e, hipSetDevice(e->device), "hipSetDevice");
    rc = upload(e, pcm, nullptr, n);
    if (rc) return rc;
    if (block_is_wide(pcm, n)) {
        rc = ensure_workspace(e, 1);
        if (rc) return rc;
        if (!e->d_wide) HIP_TRY(e, hipMalloc((void**)&e->d_wide, (size_t)11 * kMaxBlock * sizeof(int32_t)), "hipMalloc(wide residuals)");
        hipStream_t st = e->stream[0];
        HIP_TRY(e, launch_wide_block(e->d_left, n, e->cfg.zero_run_enabled ? 1 : 0, e->cfg.partitioning_enabled ? 1 : 0, e->d_wide,
                                     e->ws.plans, st), "kernel launch");
        HIP_TRY(e, hipMemcpyAsync(e->h_plans, e->ws.plans, sizeof(ChannelPlan), hipMemcpyDeviceToHost, st), "D2H plan");
        HIP_TRY(e, hipStreamSynchronize(st), "synchronize");
        return LACX_OK;
    }
    return analyze_on_device(e, e->d_left, nullptr, n, 1, 0, /*bit_depth=*/0, e->stream[0]);
}

int lacx_block_plan_only(lacx_encoder* e, const int32_t* pcm, uint32_t n, lacx_channel_plan* plan) {
    if (!e || !pcm || !plan || n == 0) return LACX_E_INVALID;
    int rc = block_size_check(e, n);
    if (rc) return rc;
    rc = block_analyze(e, pcm, n);
    if (rc) return rc;
    std::memcpy(plan, &e->h_plans[0], sizeof(ChannelPlan));
    return LACX_OK;
}

int lacx_block_encode(lacx_encoder* e, const int32_t* pcm, uint32_t n, uint8_t** out, uint64_t* out_size) {
    if (!e || !out || !out_size) return LACX_E_INVALID;
    if (n == 0) {
        // Block::Encoder::encode of an empty block: fixed-0, unpartitioned Rice with k=0, no residuals
        // (type, order, control, 7 metadata bits padded): four zero bytes.
        uint8_t* b = static_cast<uint8_t*>(std::calloc(4, 1));
        *out = b;
        *out_size = 4;
        return LACX_OK;
    }
    if (!pcm) return LACX_E_INVALID;
    int rc = block_size_check(e, n);
    if (rc) return rc;
    rc = block_analyze(e, pcm, n);
    if (rc) return rc;
    const ChannelPlan& pl = e->h_plans[0];
    // In the wide domain the reference's estimate and its emit disagree at k = 31 (the estimate drops the quotient from
    // k >= 31 on, Rice::encode from k >= 32: ref block/encoder.cpp:67-70 vs rice/rice.cpp:17-32), so the emitted size may
    // exceed the plan's there -- the bytes are the reference's either way; inside the validated domain sizes must agree.
    const bool wide = block_is_wide(pcm, n);
    if (wide && pl.payload_bytes == 0xFFFFFFFFu) return fail(e, LACX_E_RUNTIME, "encoded block is outside format limits");
    const size_t cap = (size_t)pl.payload_bytes + (wide ? (size_t)n * 8u + 64u : 0u);
    uint8_t* buf = static_cast<uint8_t*>(std::malloc(cap ? cap : 1));
    if (!buf) return fail(e, LACX_E_RUNTIME, "out of memory");
    std::vector<int32_t> scratch(n);
    const size_t wrote = emit_channel(pl, pcm, nullptr, CH_L, n, buf, cap, scratch.data());
    if (wrote == (size_t)-1 || (!wide && wrote != pl.payload_bytes)) {
        std::free(buf);
        return fail(e, LACX_E_RUNTIME, "emitted size disagrees with the device plan (internal error)");
    }
    *out = buf;
    *out_size = wrote;
    return LACX_OK;
}

int lacx_debug_stamps(unsigned long long* out32) { return debug_read_stamps(out32); }

int lacx_debug_emit_workers(lacx_encoder* e) { return e ? (int)pool_of(e).threads() : -1; }

int lacx_debug_lpc(lacx_encoder* e, const int32_t* pcm, uint32_t n, int64_t* acorr, int16_t* coef,
                   uint8_t* used) {
    if (!e || !pcm || n == 0) return LACX_E_INVALID;
    if (block_is_wide(pcm, n)) return fail(e, LACX_E_INVALID, "lacx_debug_lpc probes the streaming kernels: samples must lie in the 25-bit mid/side domain");
    int rc = block_size_check(e, n);
    if (rc) return rc;
    rc = block_analyze(e, pcm, n);
    if (rc) return rc;
    LpcSet ls;
    HIP_TRY(e, hipMemcpy(acorr, e->ws.acorr, 13 * sizeof(int64_t), hipMemcpyDeviceToHost), "D2H acorr");
    HIP_TRY(e, hipMemcpy(&ls, e->ws.lpcs, sizeof(LpcSet), hipMemcpyDeviceToHost), "D2H lpc");
    std::memcpy(coef, ls.coef, sizeof(ls.coef));
    std::memcpy(used, ls.used, sizeof(ls.used));
    return LACX_OK;
}

}  // extern "C"
