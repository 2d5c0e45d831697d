// api_fanout.cpp -- one stream over several devices, behind the C ABI ("HIP, streams, device selection and
// multi-GPU fan-out live entirely behind this shim"; 8(e): contiguous block ranges, one exchange step, host concat).
//
// The reference's LAC::Encoder::encode spreads the blocks of a stream over its worker threads itself and concatenates
// the block payloads in order (ref src/codec/lac/encoder.cpp:385-443 pool, :445-465 container).  Here the workers are
// devices: an encoder created over a device list (lacx_encoder_create_multi, or lacx_config.device = LACX_DEVICE_ALL) cuts
// the stream into contiguous block ranges [g*B/G, (g+1)*B/G), one per lane; every lane has its own host thread, its own
// encoder object (streams, workspace, pinned result region) on its device, uploads its range straight from the caller's
// buffer, and runs the same single-device pipeline as a plain encoder.  The only exchange is (payload bytes, block
// count) per lane -- RCCL all-gather of two u64 over xGMI where the devices are distinct (ncclCommInitAll, one
// communicator per lane, each lane calls ncclAllGather on its own stream), a host-side sum otherwise (RCCL refuses two
// ranks on one device; LACX_FANOUT_EXCHANGE=host|rccl forces either) -- after which every lane knows its byte offset
// and copies its payload and its slice of the block table into the final .lac in parallel.  No payload byte crosses
// xGMI.  Blocks are independent (fresh Rice state and LPC warm-up per block), so the bytes do not depend on G.
#include <dlfcn.h>

#include <atomic>
#include <barrier>
#include <condition_variable>
#include <functional>
#include <mutex>

#include "encoder_impl.h"

// ---- RCCL, loaded on first use (librccl.so is large; most processes never fan out) -----------------------------------
// Minimal declarations of the four entry points used (rccl.h:236 ncclCommInitAll, :260 ncclCommDestroy, :339
// ncclGetErrorString, :678 ncclAllGather; ncclUint64 = 5, rccl.h:464).
namespace {
using ncclComm_t = struct ncclComm*;
using ncclResult_t = int;
constexpr int kNcclUint64 = 5;
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl* rccl() {
    static std::mutex mu;
    static Rccl r;
    static bool tried = false;
    std::lock_guard<std::mutex> lock(mu);
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (r.lib) {
            r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
            r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
            r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
            r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
            if (!r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GetErrorString) r.lib = nullptr;
        }
    }
    return r.lib ? &r : nullptr;
}
}  // namespace

// One lane = one device's share of a fan-out: an encoder of its own and (lanes 1..) a host thread that runs its jobs.
struct Lane {
    lacx_encoder* enc = nullptr;  // lane 0: the owning encoder itself; others: children (owned)
    int device = 0;
    bool shares_device = false;   // another lane of the list uses the same device (rehearsal on fewer GPUs)
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, quit = false;
    // RCCL exchange buffers (device: 2 u64 in, 2 * lanes u64 out; pinned host mirrors)
    ncclComm_t comm = nullptr;
    unsigned long long* d_x = nullptr;
    unsigned long long* h_x = nullptr;
    // result of the lane's last job
    int rc = LACX_OK;
    const uint8_t* payload = nullptr;
    uint64_t pay = 0;
    const uint32_t* table = nullptr;
    uint32_t nb = 0;
    uint64_t byte_off = 0;  // of the lane's payload inside the stream's payload
    double encode_ms = 0, exchange_ms = 0, concat_ms = 0;
};

struct Fanout {
    std::vector<std::unique_ptr<Lane>> lanes;
    uint32_t min_blocks = 64;  // a lane is only used when every lane gets at least this many blocks
    bool distinct = true;      // no device twice in the list
    bool comm_tried = false, comm_ok = false;
    std::string comm_note;
    uint8_t* result = nullptr;  // the *_view result of a fanned-out call (malloc'd, kept until the next call)
    uint64_t result_cap = 0;
    lacx_fanout_stats stats{};
    lacx_timing lane0_timing{};  // lane 0 is the owning encoder: its own timing, before the merged one replaces it
};

namespace {

void lane_loop(Lane* ln) {
    for (;;) {
        std::function<void()> job;
        {
            std::unique_lock<std::mutex> lock(ln->mu);
            ln->cv.wait(lock, [&] { return ln->has_job || ln->quit; });
            if (ln->quit) return;
            job = std::move(ln->job);
        }
        job();
        {
            std::lock_guard<std::mutex> lock(ln->mu);
            ln->has_job = false;
        }
        ln->cv.notify_all();
    }
}
void lane_post(Lane* ln, std::function<void()> job) {
    {
        std::lock_guard<std::mutex> lock(ln->mu);
        ln->job = std::move(job);
        ln->has_job = true;
    }
    ln->cv.notify_all();
}
void lane_wait(Lane* ln) {
    std::unique_lock<std::mutex> lock(ln->mu);
    ln->cv.wait(lock, [&] { return !ln->has_job; });
}

// Communicators and exchange buffers, once per encoder, before the first fanned-out call that wants RCCL.
void setup_rccl(lacx_encoder* e) {
    Fanout& f = *e->fan;
    if (f.comm_tried) return;
    f.comm_tried = true;
    const uint32_t want = e->knobs.fanout_exchange;  // 0 auto, 1 host, 2 rccl
    if (want == 1u) {
        f.comm_note = "host sum (LACX_FANOUT_EXCHANGE=host)";
        return;
    }
    if (!f.distinct) {
        f.comm_note = "host sum (a device appears twice in the list: RCCL refuses two ranks on one device)";
        return;
    }
    if (f.lanes.size() < 2 && want != 2u) {
        f.comm_note = "host sum (one lane)";
        return;
    }
    Rccl* r = rccl();
    if (!r) {
        f.comm_note = "host sum (librccl.so could not be loaded)";
        return;
    }
    const int n = (int)f.lanes.size();
    std::vector<int> devs(n);
    std::vector<ncclComm_t> comms(n, nullptr);
    for (int i = 0; i < n; ++i) devs[i] = f.lanes[i]->device;
    const ncclResult_t rc = r->CommInitAll(comms.data(), n, devs.data());
    if (rc != 0) {
        f.comm_note = std::string("host sum (ncclCommInitAll: ") + r->GetErrorString(rc) + ")";
        return;
    }
    for (int i = 0; i < n; ++i) {
        Lane& ln = *f.lanes[i];
        ln.comm = comms[i];
        bool ok = hipSetDevice(ln.device) == hipSuccess;
        ok = ok && hipMalloc((void**)&ln.d_x, sizeof(unsigned long long) * 2 * (size_t)(n + 1)) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&ln.h_x, sizeof(unsigned long long) * 2 * (size_t)(n + 1), 0) == hipSuccess;
        if (!ok) {
            f.comm_note = "host sum (exchange buffers could not be allocated)";
            return;
        }
    }
    (void)hipSetDevice(e->device);
    f.comm_ok = true;
    f.comm_note = "RCCL all-gather of (payload bytes, block count) per lane";
}

// What a fanned-out call does on every lane.  encode: the lane's shard -> views into its encoder's buffers.
// place (nullable): called after the exchange with the lane's byte offset, the total payload size and the total number of
// blocks in front of the lane (the last lane to arrive at the second barrier has allocated the result by then).
struct FanCall {
    uint32_t used = 0;  // lanes that take part
    std::function<int(uint32_t lane, Lane& ln)> encode;
    std::function<void(uint64_t total_pay)> allocate;  // nullable: runs once, before any lane places its bytes
    std::function<void(uint32_t lane, Lane& ln, uint64_t blocks_before)> place;
};

int run_fanout(lacx_encoder* e, FanCall& call, const std::vector<uint64_t>& lane_frames) {
    Fanout& f = *e->fan;
    const uint32_t G = call.used;
    setup_rccl(e);
    const bool use_rccl = f.comm_ok && G == f.lanes.size();  // the communicator spans every lane of the list
    std::vector<uint64_t> pay(G, 0), nblk(G, 0);
    std::vector<int> failed(G, 0);
    uint64_t total = 0;
    bool any_failed = false;
    auto on_all_arrived = [&]() noexcept {
        total = 0;
        any_failed = false;
        for (uint32_t g = 0; g < G; ++g) {
            any_failed = any_failed || failed[g];
            f.lanes[g]->byte_off = total;
            total += pay[g];
        }
        if (!any_failed && call.allocate) call.allocate(total);
    };
    std::barrier sync((std::ptrdiff_t)G, on_all_arrived);
    auto lane_job = [&](uint32_t g) {
        Lane& ln = *f.lanes[g];
        const auto t0 = clk::now();
        ln.rc = call.encode(g, ln);
        ln.encode_ms = ms_since(t0);
        const auto t1 = clk::now();
        pay[g] = ln.rc == LACX_OK ? ln.pay : 0;
        nblk[g] = ln.rc == LACX_OK ? ln.nb : 0;
        failed[g] = ln.rc != LACX_OK;
        if (use_rccl) {
            // every lane takes part whether or not its shard failed: (all ones, 0) marks a failed lane
            Rccl* r = rccl();
            hipStream_t s = ln.enc->stream[0];
            ln.h_x[0] = failed[g] ? ~0ull : pay[g];
            ln.h_x[1] = nblk[g];
            bool ok = hipSetDevice(ln.device) == hipSuccess;
            ok = ok && hipMemcpyAsync(ln.d_x, ln.h_x, 2 * sizeof(unsigned long long), hipMemcpyHostToDevice, s) == hipSuccess;
            ok = ok && r->AllGather(ln.d_x, ln.d_x + 2, 2, kNcclUint64, ln.comm, s) == 0;
            ok = ok && hipMemcpyAsync(ln.h_x + 2, ln.d_x + 2, 2 * (size_t)G * sizeof(unsigned long long), hipMemcpyDeviceToHost, s) == hipSuccess;
            ok = ok && hipStreamSynchronize(s) == hipSuccess;
            if (ok) {
                // the lane's own view of everybody's sizes: it must agree with what the host barrier sums below
                uint64_t before = 0;
                for (uint32_t k = 0; k < g; ++k) before += ln.h_x[2 + 2 * k] == ~0ull ? 0 : ln.h_x[2 + 2 * k];
                ln.byte_off = before;
            } else if (ln.rc == LACX_OK) {
                ln.rc = fail(ln.enc, LACX_E_DEVICE, "RCCL all-gather of the shard sizes failed");
                failed[g] = 1;
            }
        }
        const uint64_t rccl_off = ln.byte_off;
        sync.arrive_and_wait();  // every lane's sizes are known; the completion step has summed them and allocated the result
        if (use_rccl && !any_failed && rccl_off != ln.byte_off) {
            ln.rc = fail(ln.enc, LACX_E_RUNTIME, "shard offsets from the RCCL exchange disagree with the host's (internal error)");
        }
        ln.exchange_ms = ms_since(t1);
        const auto t2 = clk::now();
        if (!any_failed && ln.rc == LACX_OK && call.place) {
            uint64_t before = 0;
            for (uint32_t k = 0; k < g; ++k) before += nblk[k];
            call.place(g, ln, before);
        }
        ln.concat_ms = ms_since(t2);
    };
    for (uint32_t g = 1; g < G; ++g) lane_post(f.lanes[g].get(), [&lane_job, g] { lane_job(g); });
    lane_job(0);
    for (uint32_t g = 1; g < G; ++g) lane_wait(f.lanes[g].get());
    (void)hipSetDevice(e->device);
    // statistics of the call
    lacx_fanout_stats& st = f.stats;
    st = lacx_fanout_stats{};
    st.lanes_used = G;
    st.exchange = use_rccl ? LACX_EXCHANGE_RCCL : LACX_EXCHANGE_HOST;
    for (uint32_t g = 0; g < G && g < LACX_MAX_FANOUT; ++g) {
        const Lane& ln = *f.lanes[g];
        st.device[g] = ln.device;
        st.blocks[g] = (uint32_t)nblk[g];
        st.lane_frames[g] = lane_frames[g];
        st.payload_bytes[g] = pay[g];
        st.encode_ms[g] = ln.encode_ms;
        st.exchange_ms = std::max(st.exchange_ms, ln.exchange_ms);
        st.concat_ms = std::max(st.concat_ms, ln.concat_ms);
    }
    // errors: the reference reports the first bad sample of the LEFT channel anywhere in the stream, then of the right
    // (ref lac/encoder.cpp:238-241); every lane reports its own shard that way, so the lowest shard with a left error wins,
    // else the lowest with a right error; any other failure: the lowest lane's.
    for (int pass = 0; pass < 2; ++pass) {
        uint64_t f0 = 0;
        for (uint32_t g = 0; g < G; ++g) {
            const Lane& ln = *f.lanes[g];
            if (ln.rc == LACX_E_INVALID && ln.enc->bad_channel == pass) {
                return fail(e, LACX_E_INVALID, std::string(pass ? "right" : "left") + " sample at index " +
                                                   std::to_string(f0 + ln.enc->bad_index) + " is outside the configured PCM bit depth");
            }
            f0 += lane_frames[g];
        }
    }
    for (uint32_t g = 0; g < G; ++g) {
        const Lane& ln = *f.lanes[g];
        if (ln.rc != LACX_OK) return g == 0 ? ln.rc : fail(e, ln.rc, ln.enc->err);
    }
    return LACX_OK;
}

// Lanes a stream of nb blocks is spread over: every lane of the list, unless that leaves a lane with fewer than
// min_blocks blocks (the reference uses min(threads, blocks) workers, ref lac/encoder.cpp:385-390).
uint32_t lanes_for(const Fanout& f, uint32_t nb) {
    const uint32_t by_size = std::max<uint32_t>(1u, nb / std::max<uint32_t>(1u, f.min_blocks));
    return std::max<uint32_t>(1u, std::min<uint32_t>((uint32_t)f.lanes.size(), std::min(by_size, nb)));
}

// Timing of the owning encoder after a fanned-out call: wall clock of the call, per-phase maxima over the lanes.
void merge_timing(lacx_encoder* e, uint32_t G, clk::time_point t0) {
    lacx_timing t{};
    for (uint32_t g = 0; g < G; ++g) {
        const lacx_timing& c = e->fan->lanes[g]->enc->timing;
        t.h2d_ms = std::max(t.h2d_ms, c.h2d_ms);
        t.analysis_ms = std::max(t.analysis_ms, c.analysis_ms);
        t.ingest_ms = std::max(t.ingest_ms, c.ingest_ms);
        t.probe_ms = std::max(t.probe_ms, c.probe_ms);
        t.full_ms = std::max(t.full_ms, c.full_ms);
        t.d2h_ms = std::max(t.d2h_ms, c.d2h_ms);
        t.emit_ms = std::max(t.emit_ms, c.emit_ms);
        t.full_exec_ms = std::max(t.full_exec_ms, c.full_exec_ms);
        t.full_slots += c.full_slots;
        t.probe_slots += c.probe_slots;
        t.full_launches = std::max(t.full_launches, c.full_launches);
        t.regrows += c.regrows;
        t.emit_direct += c.emit_direct;
        t.moved_by_k_pack += c.moved_by_k_pack;
        t.silent_copies += c.silent_copies;
        t.packer_gave_up += c.packer_gave_up;
    }
    t.total_ms = ms_since(t0);
    e->timing = t;
}

}  // namespace

namespace lacx_host {

bool is_fanout(const lacx_encoder* e) { return e && e->fan; }

void destroy_fanout(lacx_encoder* e) {
    if (!e->fan) return;
    Fanout* f = e->fan;
    for (size_t g = 0; g < f->lanes.size(); ++g) {
        Lane& ln = *f->lanes[g];
        if (ln.th.joinable()) {
            {
                std::lock_guard<std::mutex> lock(ln.mu);
                ln.quit = true;
            }
            ln.cv.notify_all();
            ln.th.join();
        }
        if (ln.comm) {
            if (Rccl* r = rccl()) (void)r->CommDestroy(ln.comm);
        }
        if (ln.d_x || ln.h_x) {
            (void)hipSetDevice(ln.device);
            if (ln.d_x) (void)hipFree(ln.d_x);
            if (ln.h_x) (void)hipHostFree(ln.h_x);
        }
        if (g > 0 && ln.enc) lacx_encoder_destroy(ln.enc);
    }
    std::free(f->result);
    delete f;
    e->fan = nullptr;
}

// Whole stream in host memory -> complete .lac, over the lanes.  layout 0: planar int32 (hs.p0 = left, hs.p1 = right or
// null); 1 / 2: the WAV data chunk (interleaved int16 / packed int24).  *out: malloc'd (owned == true) or the encoder's
// own result buffer (valid until its next call).
int fanout_encode_host(lacx_encoder* e, const HostSrc& hs, int layout, int channels, uint64_t frames, bool owned,
                       uint8_t** out, uint64_t* out_size) {
    Fanout& f = *e->fan;
    const auto t0 = clk::now();
    const uint32_t nb = blocks_for(frames);
    const uint32_t G = lanes_for(f, nb);
    const uint64_t head = 10 + 4 + 8ull * nb;
    uint8_t* buf = nullptr;
    std::vector<uint64_t> lane_f0(G), lane_frames(G);
    for (uint32_t g = 0; g < G; ++g) {
        uint32_t b0 = 0, cnt = 0;
        lacx_fanout_range(nb, G, g, &b0, &cnt);
        lane_f0[g] = (uint64_t)b0 * kMaxBlock;
        lane_frames[g] = std::min<uint64_t>(frames, (uint64_t)(b0 + cnt) * kMaxBlock) - lane_f0[g];
    }
    FanCall call;
    call.used = G;
    call.encode = [&](uint32_t g, Lane& ln) -> int {
        HostSrc mine = hs;
        mine.p0 = hs.p0 + lane_f0[g] * hs.frame_bytes;
        if (hs.p1) mine.p1 = hs.p1 + lane_f0[g] * hs.frame_bytes;
        ln.enc->timing = lacx_timing{};
        return encode_host_shard_view(ln.enc, mine, layout, channels, lane_frames[g], &ln.payload, &ln.pay, &ln.table, &ln.nb);
    };
    bool oom = false;
    call.allocate = [&](uint64_t total) {
        const uint64_t need = head + total;
        if (owned) {
            buf = static_cast<uint8_t*>(std::malloc(need ? need : 1));
        } else {
            if (need > f.result_cap) {
                std::free(f.result);
                f.result = static_cast<uint8_t*>(std::malloc(need + need / 8 + 4096));
                f.result_cap = f.result ? need + need / 8 + 4096 : 0;
            }
            buf = f.result;
        }
        oom = buf == nullptr;
        if (buf) {
            write_frame_header(stream_params(e->cfg, channels), buf);
            put32(buf + 10, nb);
        }
    };
    std::atomic<bool> bad_block{false};
    call.place = [&](uint32_t, Lane& ln, uint64_t blocks_before) {
        if (!buf) return;
        for (uint32_t b = 0; b < ln.nb; ++b) {
            if (ln.table[2 * b + 1] == 0) bad_block = true;
            put32(buf + 14 + 8ull * (blocks_before + b), ln.table[2 * b]);
            put32(buf + 18 + 8ull * (blocks_before + b), ln.table[2 * b + 1]);
        }
        // (one thread per lane copies its own payload: the concat runs G-wide; pieces above 32 MB split further)
        if (ln.pay > (32ull << 20)) big_copy(buf + head + ln.byte_off, ln.payload, ln.pay);
        else std::memcpy(buf + head + ln.byte_off, ln.payload, ln.pay);
    };
    const int rc = run_fanout(e, call, lane_frames);
    f.lane0_timing = e->timing;
    merge_timing(e, G, t0);
    if (rc != LACX_OK) {
        if (owned) std::free(buf);
        return rc;
    }
    if (oom) return fail(e, LACX_E_RUNTIME, "out of memory");
    if (bad_block) {
        if (owned) std::free(buf);
        return fail(e, LACX_E_RUNTIME, "encoded block size is outside format limits");
    }
    uint64_t total = 0;
    for (uint32_t g = 0; g < G; ++g) total += f.lanes[g]->pay;
    *out = buf;
    *out_size = head + total;
    e->timing.total_ms = ms_since(t0);
    return LACX_OK;
}

// Shards already resident in device memory, shard g on lane g's device (the bench's timed region; a caller that produces
// the PCM on the devices): every lane encodes its shard, the sizes are exchanged, out[g] views the lane's payload and
// block table in its pinned result region together with its byte offset in the stream's payload.  No concatenation
// (lacx_assemble does that from the views when a contiguous .lac is wanted).
int fanout_encode_resident(lacx_encoder* e, const lacx_fanout_shard* shards, uint32_t n, lacx_fanout_out* out) {
    Fanout& f = *e->fan;
    const auto t0 = clk::now();
    if (n == 0 || n > f.lanes.size()) return fail(e, LACX_E_INVALID, "more shards than the encoder has lanes");
    std::vector<uint64_t> lane_frames(n);
    for (uint32_t g = 0; g < n; ++g) {
        lane_frames[g] = shards[g].frames;
        // every shard but the last ends on a block boundary (it is a block range of one stream)
        if (g + 1 < n && shards[g].frames % (uint64_t)kMaxBlock != 0)
            return fail(e, LACX_E_INVALID, "shard " + std::to_string(g) + " does not end on a block boundary");
    }
    FanCall call;
    call.used = n;
    call.encode = [&](uint32_t g, Lane& ln) -> int {
        lacx_encoder* c = ln.enc;
        int rc = lacx_encode_shard_pcm_device_begin(c, &shards[g].pcm, shards[g].frames, nullptr);
        if (rc) return rc;
        return lacx_encode_shard_end(c, &ln.payload, &ln.pay, &ln.table, &ln.nb);
    };
    const int rc = run_fanout(e, call, lane_frames);
    // (lane 0 is this encoder: its own timing was set by the shard call; keep it where the statistics can find it)
    f.lane0_timing = e->timing;
    merge_timing(e, n, t0);
    if (rc != LACX_OK) return rc;
    for (uint32_t g = 0; g < n; ++g) {
        const Lane& ln = *f.lanes[g];
        out[g].payload = ln.payload;
        out[g].payload_size = ln.pay;
        out[g].table = ln.table;
        out[g].nblocks = ln.nb;
        out[g].device = ln.device;
        out[g].byte_offset = ln.byte_off;
    }
    return LACX_OK;
}

}  // namespace lacx_host

extern "C" {

void lacx_fanout_range(uint32_t nblocks, uint32_t nlanes, uint32_t lane, uint32_t* first, uint32_t* count) {
    if (nlanes == 0) nlanes = 1;
    const uint64_t b0 = (uint64_t)lane * nblocks / nlanes, b1 = (uint64_t)(lane + 1) * nblocks / nlanes;
    if (first) *first = (uint32_t)b0;
    if (count) *count = (uint32_t)(b1 - b0);
}

int lacx_encoder_create_multi(const lacx_config* cfg, const int32_t* devices, uint32_t ndevices, uint32_t min_blocks_per_device,
                              lacx_encoder** out) {
    if (!cfg || !out || !devices || ndevices == 0 || ndevices > LACX_MAX_FANOUT) return LACX_E_INVALID;
    for (uint32_t i = 0; i < ndevices; ++i)
        if (devices[i] < 0) return LACX_E_INVALID;
    lacx_config c0 = *cfg;
    c0.device = devices[0];
    lacx_encoder* e = nullptr;
    int rc = lacx_encoder_create(&c0, &e);
    if (rc) return rc;
    // (a list of one device is a plain encoder, unless the RCCL exchange is forced: then the whole fan-out machinery runs
    // with one lane -- how a one-GPU box exercises the communicator set-up and the all-gather)
    if (ndevices > 1 || e->knobs.fanout_exchange == 2u) {
        Fanout* f = new Fanout();
        if (min_blocks_per_device) f->min_blocks = min_blocks_per_device;
        for (uint32_t i = 0; i < ndevices; ++i) {
            auto ln = std::make_unique<Lane>();
            ln->device = devices[i];
            for (uint32_t k = 0; k < ndevices; ++k)
                if (k != i && devices[k] == devices[i]) ln->shares_device = true;
            f->distinct = f->distinct && !ln->shares_device;
            if (i == 0) {
                ln->enc = e;
            } else {
                lacx_config ci = *cfg;
                ci.device = devices[i];
                rc = lacx_encoder_create(&ci, &ln->enc);
                if (rc) break;
                ln->th = std::thread(lane_loop, ln.get());
            }
            // Lanes that share a device (a rehearsal on fewer GPUs than lanes): persistent analysis workgroups never retire,
            // so two such grids and their packers on one device starve each other -- the launched grid time-slices.
            if (ln->shares_device) ln->enc->knobs.persistent = false;
            f->lanes.push_back(std::move(ln));
        }
        e->fan = f;
        if (rc) {
            lacx_encoder_destroy(e);
            return rc;
        }
    }
    *out = e;
    return LACX_OK;
}

uint32_t lacx_encoder_lanes(const lacx_encoder* e) { return e && e->fan ? (uint32_t)e->fan->lanes.size() : (e ? 1u : 0u); }

int lacx_encode_fanout_resident(lacx_encoder* e, const lacx_fanout_shard* shards, uint32_t nshards, lacx_fanout_out* out) {
    if (!e || !shards || !out) return LACX_E_INVALID;
    if (!e->fan) {  // a plain encoder: one shard, no exchange
        if (nshards != 1) return fail(e, LACX_E_INVALID, "more shards than the encoder has lanes");
        const int rc = lacx_encode_shard_pcm_device_view(e, &shards[0].pcm, shards[0].frames, nullptr, &out[0].payload,
                                                         &out[0].payload_size, &out[0].table, &out[0].nblocks);
        if (rc) return rc;
        out[0].device = e->device;
        out[0].byte_offset = 0;
        return LACX_OK;
    }
    if (e->cfg.flags & LACX_FLAG_HOST_EMIT)
        return fail(e, LACX_E_INVALID, "this entry point needs the device-side emit (LACX_FLAG_HOST_EMIT is set)");
    return fanout_encode_resident(e, shards, nshards, out);
}

int lacx_get_fanout_stats(const lacx_encoder* e, lacx_fanout_stats* out) {
    if (!e || !out) return LACX_E_INVALID;
    if (!e->fan) {
        *out = lacx_fanout_stats{};
        out->lanes_used = 1;
        out->device[0] = e->device;
        return LACX_OK;
    }
    *out = e->fan->stats;
    return LACX_OK;
}

const char* lacx_fanout_exchange_note(const lacx_encoder* e) { return e && e->fan ? e->fan->comm_note.c_str() : ""; }

int lacx_get_lane_timing(const lacx_encoder* e, uint32_t lane, lacx_timing* out) {
    if (!e || !out) return LACX_E_INVALID;
    if (!e->fan) {
        if (lane != 0) return LACX_E_INVALID;
        *out = e->timing;
        return LACX_OK;
    }
    if (lane >= e->fan->lanes.size()) return LACX_E_INVALID;
    *out = lane == 0 ? e->fan->lane0_timing : e->fan->lanes[lane]->enc->timing;
    return LACX_OK;
}

}  // extern "C"
