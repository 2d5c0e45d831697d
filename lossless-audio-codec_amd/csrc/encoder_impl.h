// encoder_impl.h -- what the host-side translation units of liblacx.so share: the encoder object behind the opaque
// lacx_encoder handle and the internal functions of api_core.cpp (device, workspace, helpers) and api_pipeline.cpp (the
// launch pipelines).  Not part of the C ABI (include/lacx.h).
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
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

namespace lacx_host {
constexpr int kStreams = 4;
constexpr int kMaxChunks = 16;
constexpr uint32_t kMinChunkBlocks = 192;  // >= 1.5 rounds of 1024-thread workgroups over 256 CUs
using clk = std::chrono::steady_clock;
struct Chunk {
    uint32_t first, count;
};

// Host-resident input of an encode whose upload is pipelined with the analysis: chunk c's PCM is copied to the device
// on chunk c's stream right in front of its kernels, so the upload of chunk c+1 overlaps the analysis of chunk c
// (ref src/main.cpp:658-675 reads the whole file first, then encodes).
struct HostSrc {
    const uint8_t* p0 = nullptr;  // planar: left; interleaved: the WAV data chunk
    const uint8_t* p1 = nullptr;  // planar: right (null for mono)
    uint64_t frame_bytes = 0;     // bytes per frame in p0 (and p1)
};
}  // namespace lacx_host
using namespace lacx_host;

// Every environment knob of the encode path, read ONCE when the encoder is created (lacx_encoder_create): no entry
// point reads the environment afterwards.  All are tuning / diagnostic switches; none changes the bytes produced.
struct Knobs {
    bool stream_priority = true;   // LACX_NO_STREAM_PRIORITY unset
    // LACX_STREAM_PRIO=main,chunks,pack: priority level (-1 high, 0 normal, 1 low) of the first analysis stream, of the
    // later pipeline chunks' streams and of the streaming packer's stream.  The packer's level must be one that NOTHING
    // else uses: hardware queues are pooled per level (see ensure_device).
    int prio_main = -1, prio_chunks = -1, prio_pack = 1;
    bool fused_emit = true;        // LACX_FUSED_EMIT != 0
    bool emit_staged = false;      // LACX_EMIT_STAGED
    bool direct_packer = false;    // LACX_DIRECT_PACKER: the packer stores into pinned host memory itself (round-2 layout)
    bool packer = true;            // LACX_NO_PACKER unset
    bool chain = true;             // LACX_NO_CHAIN unset
    bool persistent = true;        // LACX_NO_PERSISTENT unset: whole-block analysis as persistent workgroups
    bool debug_drain = false;      // LACX_DEBUG_DRAIN
    bool two_copy_streams = true;  // LACX_ONE_COPY_STREAM unset
    uint64_t pinned_cap_bytes = 0; // LACX_PINNED_CAP_BYTES (tests force the regrow path with it)
    uint32_t debug_skip = 0;       // LACX_DEBUG_SKIP: test hooks (bits 10, 11, 13) / ablations; only honoured by a library
                                   // built with -DLACX_TEST_HOOKS (liblacx_hooks.so)
    uint32_t pipe_chunks = 0;      // LACX_PIPE_CHUNKS
    std::string pipe_split;        // LACX_PIPE_SPLIT
    uint32_t drain_fence = 0;      // LACX_DRAIN_FENCE
    bool lazy_repair = true;       // LACX_NO_LAZY_REPAIR unset
    bool silent_template = true;   // LACX_NO_SILENT_TEMPLATE unset: silent slots after the first are copies (kernels.h)
    bool front_halves = true;      // LACX_NO_FRONT_HALVES unset: a one-chunk shard's front kernels in two block halves on two streams
    bool front_stream_split = false; // LACX_FRONT_STREAM: upload pipeline with the front kernels on a high-priority stream (experiment)
    uint32_t fanout_exchange = 0;  // LACX_FANOUT_EXCHANGE: 0 auto (RCCL where the devices are distinct), 1 host, 2 rccl
    LaunchTuning tune;             // LACX_PERSISTENT_GRID, LACX_PACK_NAP, LACX_PACK_GRID
};

// One persistent host thread per encoder that performs the host-to-device copies of an encode whose input starts in host
// memory.  hipMemcpyAsync from pageable memory returns only when the copy is (all but) done, so issued from the calling
// thread every chunk's upload stands between that thread and the next chunk's kernel launches; issued from here, the
// calling thread enqueues chunk c's kernels the moment chunk c's bytes are on their way, while chunk c + 1 is uploading.
class Uploader {
public:
    Uploader() : th_([this] { loop(); }) {}
    ~Uploader() {
        {
            std::lock_guard<std::mutex> lock(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void post(std::function<void()> job) {
        {
            std::lock_guard<std::mutex> lock(mu_);
            job_ = std::move(job);
            busy_ = true;
        }
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [&] { return !busy_; });
    }

private:
    void loop() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lock(mu_);
                cv_.wait(lock, [&] { return busy_ || quit_; });
                if (quit_) return;
                job = std::move(job_);
            }
            job();
            {
                std::lock_guard<std::mutex> lock(mu_);
                busy_ = false;
            }
            cv_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool busy_ = false, quit_ = false;
    std::thread th_;
};

struct Fanout;  // api_fanout.cpp: the lanes of an encoder that spreads a stream over several devices

struct lacx_encoder {
    lacx_config cfg{};
    Knobs knobs{};
    Fanout* fan = nullptr;  // non-null: lacx_encode / lacx_encode_wav* split the stream into block ranges over fan's lanes
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
        bool lazy_repair = false;  // k_pack / k_emit not enqueued: the gather kernel says whether they are needed
        uint32_t fuse_items = 0;
        uint64_t emit_cap = 0;
        uint8_t* emit_dst = nullptr;
        bool front_split = false;  // upload pipeline: chunk c ran on stream 1 + c % 3
        uint32_t next_range = 0;  // copy-engine drain: first range not yet looked at, bytes already on their way
        uint64_t drained_to = 0;
        hipStream_t st[4] = {};
        clk::time_point t0;
        // inputs of the call, kept for the re-emit after a too-small result reservation
        const int32_t* d_left = nullptr;
        const int32_t* d_right = nullptr;
        uint64_t frames = 0;
        int layout = 0;
    } pend;
    hipEvent_t prologue = nullptr;  // per-call memsets done (the chunk streams wait for it)
    hipEvent_t aux_ev[2] = {};      // front kernels in two halves (LaunchTuning::aux_ev)
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
    unsigned long long* h_sizes = nullptr;  // pinned copy of the size records (lazy repair: the host builds the block table from them)
    uint32_t h_sizes_cap = 0;
    uint32_t* h_emitted = nullptr; // pinned copy of ws.emitted (statistics of the fused emit)
    uint32_t h_emitted_cap = 0;
    unsigned long long* d_tspan = nullptr;  // [2][kMaxChunks]: ~first-start / last-end device clock of k_analyze<16,1024>
    SilentTemplate* d_silent = nullptr;  // zeroed per call: the finished channel block of a silent slot (kernels.h)
    uint32_t* d_work_ctr = nullptr;  // zeroed per call: work counters of the persistent analysis, 8 per pipeline chunk
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
    // host-resident input: the uploader thread, its stream, one event per pipeline chunk, and the hand-over words
    std::unique_ptr<Uploader> uploader;
    hipStream_t up_stream = nullptr;
    hipStream_t front_stream = nullptr;  // upload pipeline: every chunk's front kernels, above the chunk streams in priority
    hipEvent_t up_ev[kMaxChunks] = {};
    std::atomic<int> up_done[kMaxChunks] = {};  // 1: the chunk's copy has been issued and its event recorded; -1: failed
    double up_ms = 0;                           // host time the uploader spent in the copies of the call
    std::string err;
    int bad_channel = -1;    // last sample-range error: 0 left / 1 right, and the sample's index in the call's input
    uint64_t bad_index = 0;  // (the fan-out rebuilds the reference's message with the stream-wide index)
    lacx_timing timing{};
};

#define HIP_TRY(e, call, what)                                  \
    do {                                                        \
        const hipError_t _err = (call);                         \
        if (_err != hipSuccess) return hip_fail(e, _err, what); \
    } while (0)

namespace lacx_host {
// api_core.cpp and api_pipeline.cpp
double ms_since(clk::time_point t0);
int fail(lacx_encoder* e, int code, const std::string& msg);
int hip_fail(lacx_encoder* e, hipError_t err, const char* what);
void big_copy(uint8_t* dst, const uint8_t* src, uint64_t n);
int ensure_device(lacx_encoder* e);
EmitPool& pool_of(lacx_encoder* e);
void free_workspace(lacx_encoder* e);
int ensure_workspace(lacx_encoder* e, uint32_t nblocks);
int ensure_slots(lacx_encoder* e, uint32_t nblocks, int channels, int bit_depth = 0);
int ensure_pcm(lacx_encoder* e, uint64_t frames, bool stereo);
bool rate_ok(uint32_t sr);
int validate_stream_args(lacx_encoder* e, const void* left, uint64_t frames);
uint32_t blocks_for(uint64_t frames);
AnalyzeParams make_params(const lacx_encoder* e, uint64_t frames, int channels, int stereo_mode, int bit_depth, int layout = 0);
LaunchSet one_stream_set(const AnalyzeParams& prm, const int32_t* left, const int32_t* right, uint32_t fuse_items = 0, uint64_t out_cap = 0);
const LaunchSet& bind(LaunchSet& ls);
DeviceWorkspace ws_at(const DeviceWorkspace& ws, uint32_t first_block);
std::vector<Chunk> plan_chunks(const Knobs& kn, uint32_t nb, bool device_emit = false, bool fused = false, bool upload = false);
Knobs read_knobs();
void add_chunk_timing(lacx_encoder* e, int c);
void count_slots(lacx_encoder* e, uint32_t first, uint32_t count);
int enqueue_chunk(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, uint64_t frames, int channels, int stereo_mode, int bit_depth, const Chunk& ck, int c, hipStream_t st);
int check_sample_range(lacx_encoder* e, uint32_t nb);
void reset_device_timing(lacx_encoder* e);
int analyze_on_device(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, uint64_t frames, int channels, int stereo_mode, int bit_depth, hipStream_t st);
StreamParams stream_params(const lacx_config& c, int channels);
void put32(uint8_t* p, uint32_t v);
uint64_t payload_upper_bound(uint64_t frames, int channels, uint32_t nb);
int encode_pipelined(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, const int32_t* h_left, const int32_t* h_right, uint64_t frames, hipStream_t user_stream, uint64_t head, uint8_t** buf_out, uint64_t* payload_size, std::vector<uint64_t>& offsets);
uint64_t pinned_reservation(const lacx_encoder* e, uint64_t frames, int channels, uint32_t nb);
int encode_device_begin(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, uint64_t frames, hipStream_t user_stream, int layout = 0, int layout_channels = 0, const HostSrc* hs = nullptr);
int encode_device_begin_impl(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, uint64_t frames, hipStream_t user_stream, int layout, int layout_channels, const HostSrc* hs);
int reemit_into_regrown_buffer(lacx_encoder* e, uint64_t* payload_size);
void drain_pump(lacx_encoder* e);
int encode_device_end(lacx_encoder* e, uint64_t* payload_size);
int encode_pipelined_device(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, uint64_t frames, hipStream_t user_stream, uint64_t* payload_size, int layout = 0, int layout_channels = 0, const HostSrc* hs = nullptr);
int fetch_pcm_if_needed(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, uint64_t frames, const int32_t*& h_left, const int32_t*& h_right, std::vector<int32_t>& tl, std::vector<int32_t>& tr);
int upload(lacx_encoder* e, const int32_t* left, const int32_t* right, uint64_t frames);
int prepare(lacx_encoder* e, const void* left, uint64_t frames);
int encode_batch(lacx_encoder* e, const lacx_batch_item* items, uint32_t n, hipStream_t user_stream, lacx_batch_out* out,
                 const std::vector<uint64_t>* exact_caps = nullptr);
int fill_table(lacx_encoder* e, uint8_t* buf, uint32_t nb, const std::vector<uint64_t>& offsets);
// api_encode.cpp: `frames` frames of host PCM (layout 0: planar int32 in hs.p0 / hs.p1; 1 / 2: the WAV data chunk) on e's
// device; the results are views into e's buffers (valid until its next call)
int encode_host_shard_view(lacx_encoder* e, const HostSrc& hs, int layout, int channels, uint64_t frames, const uint8_t** payload,
                           uint64_t* payload_size, const uint32_t** table, uint32_t* nblocks);
// api_fanout.cpp
bool is_fanout(const lacx_encoder* e);
void destroy_fanout(lacx_encoder* e);
int fanout_encode_host(lacx_encoder* e, const HostSrc& hs, int layout, int channels, uint64_t frames, bool owned, uint8_t** out,
                       uint64_t* out_size);
}  // namespace lacx_host
