// emit.cpp -- host emit of the LAC channel-block bitstream from device-produced plans.
//
// Bitstream layout per channel block (ref src/codec/block/encoder.cpp:780-822, docs/format.md):
//   u8 predictor_type | u8 order | i16 coeff[1..order] (LPC) | u8 control |
//   (u2 mode, u5 k) x partitions | residual tokens, MSB first | zero pad to a byte.
// The four residual token grammars follow encoder.cpp:585-771; the adaptive Rice parameter follows
// Rice::adapt_k (src/codec/rice/rice.hpp:45-114) when the block is unpartitioned and
// adapt_k_stateless (encoder.cpp:72-77) inside partitions (encoder.cpp:556).
#include "emit.h"

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

#include "analyze_core.h"  // kmean(), zigzag32(), slot_fetch()

namespace lacx {
namespace {

// MSB-first bit sink over a caller-provided buffer (semantics of ref bitstream/bit_writer.cpp).
class BitSink {
public:
    BitSink(uint8_t* out, size_t cap) : p_(out), end_(out + cap), acc_(0), nbits_(0), overflow_(false) {}

    inline void put(uint32_t value, unsigned n) {  // n in 1..32, value < 2^n
        acc_ = (acc_ << n) | value;
        nbits_ += n;
        if (nbits_ >= 32) drain();
    }
    inline void ones(uint32_t q) {
        while (q >= 32) {
            put(0xFFFFFFFFu, 32);
            q -= 32;
        }
        if (q) put((1u << q) - 1u, q);
    }
    // Rice code of u with parameter k: q ones, a zero, k remainder bits (ref rice.cpp:17-32).
    inline void rice(uint32_t u, uint32_t k) {
        const uint32_t q = (k >= 32u) ? 0u : (u >> k);
        if (q < 32u && q + 1u + k <= 32u) {  // (q itself can be anything up to 2^32 - 1 outside the validated domain)
            const uint32_t rem = k ? (u & ((1u << k) - 1u)) : 0u;
            const uint32_t un = (q ? (((1u << q) - 1u) << 1) : 0u);  // q ones then a zero
            put(k ? ((un << k) | rem) : un, q + 1u + k);
            return;
        }
        ones(q);
        put(0, 1);
        if (k) put(u & ((k >= 32u) ? 0xFFFFFFFFu : ((1u << k) - 1u)), k);
    }
    size_t finish() {  // zero pad to a byte; returns bytes written or (size_t)-1 on overflow
        while (nbits_ >= 8) {
            push((uint8_t)(acc_ >> (nbits_ - 8)));
            nbits_ -= 8;
        }
        if (nbits_) {
            push((uint8_t)((acc_ << (8 - nbits_)) & 0xFFu));
            nbits_ = 0;
        }
        return overflow_ ? (size_t)-1 : (size_t)(p_ - start());
    }
    void set_start(uint8_t* s) { start_ = s; }
    uint8_t* start() const { return start_; }

private:
    inline void push(uint8_t b) {
        if (p_ < end_) {
            *p_++ = b;
        } else {
            overflow_ = true;
        }
    }
    inline void drain() {
        while (nbits_ >= 8) {
            push((uint8_t)(acc_ >> (nbits_ - 8)));
            nbits_ -= 8;
        }
    }
    uint8_t* p_;
    uint8_t* end_;
    uint8_t* start_ = nullptr;
    uint64_t acc_;
    unsigned nbits_;
    bool overflow_;
};

// Stateful adaptive-k model, i.e. the value Rice::adapt_k returns after each sample
// (ref rice.hpp:45-114), kept as running window sums instead of the reference's state struct.
class DriftModel {
public:
    DriftModel() { std::memset(this, 0, sizeof(*this)); }
    inline uint32_t push(uint32_t u) {
        sum_ += u;
        const uint32_t c = ++count_;
        const uint32_t slot = (c - 1u) & 255u;
        if (c > 256u) wsum_ -= ring_[slot];
        ring_[slot] = u;
        wsum_ += u;
        uint32_t km = kmean(sum_, c);
        if (km > 31u) km = 31u;  // ref rice.hpp:68-71 (only a mean of 2^31 and more gets here: Block::Encoder's wide domain)
        const uint32_t q = (km >= 31u) ? 0u : (u >> km);
        const uint32_t mslot = (c - 1u) % 96u;
        large_ += (q > 3u) - fl_[mslot];
        zero_ += (q == 0u) - fz_[mslot];
        fl_[mslot] = q > 3u;
        fz_[mslot] = q == 0u;
        int bias = 0;
        const uint64_t X = sum_ + (c >> 1);
        if (c > 256u && X >= c) {  // below 257 samples the local mean equals the global mean
            const uint64_t L = (wsum_ + 128u) >> 8;
            const uint64_t U = (3u * L + 3u) >> 2;          // mean < ceil(3L/4)   <=> 3L > 4 mean
            const uint64_t D = L + (L + 3u) / 3u + 1u;      // mean >= (4L+3)/3 + 1 <=> 4L+3 < 3 mean
            if (X < U * c) {
                bias = 1;
            } else if (X >= D * c) {
                bias = -1;
            }
        }
        if (c >= 96u) {
            if (large_ * 4u >= 288u) {
                bias = bias + 1 < 1 ? bias + 1 : 1;
            } else if (zero_ * 5u >= 384u) {
                bias = bias - 1 > -1 ? bias - 1 : -1;
            }
        }
        int k = (int)km + bias;
        return (uint32_t)(k < 0 ? 0 : (k > 31 ? 31 : k));
    }

private:
    uint64_t sum_, wsum_;
    uint32_t count_, large_, zero_;
    uint32_t ring_[256];
    uint8_t fl_[96], fz_[96];
};

struct MeanModel {  // adapt_k_stateless
    uint64_t sum = 0;
    uint32_t count = 0;
    inline uint32_t push(uint32_t u) {
        sum += u;
        ++count;
        const uint32_t k = kmean(sum, count);
        return k > 31u ? 31u : k;  // ref block/encoder.cpp:72-77
    }
    inline uint32_t skip(uint32_t run) {
        count += run;
        const uint32_t k = kmean(sum, count);
        return k > 31u ? 31u : k;
    }
};

void compute_residual(const ChannelPlan& plan, const int32_t* a, const int32_t* b, int kind, uint32_t n,
                      int32_t* res) {
    SlotSrc src{a, b, kind};
    // materialise the channel first (M/S derive from L/R), then predict in place from the back
    for (uint32_t i = 0; i < n; ++i) res[i] = slot_fetch(src, (int64_t)i);
    const int order = plan.order;
    if (plan.predictor_type == 0) {
        if (order == 0) return;
        for (uint32_t i = n; i-- > (uint32_t)order;) {
            int64_t pred = 0;
            switch (order) {
                case 1: pred = res[i - 1]; break;
                case 2: pred = 2LL * res[i - 1] - res[i - 2]; break;
                case 3: pred = 3LL * res[i - 1] - 3LL * res[i - 2] + res[i - 3]; break;
                default: pred = 4LL * res[i - 1] - 6LL * res[i - 2] + 4LL * res[i - 3] - res[i - 4]; break;
            }
            res[i] = (int32_t)((int64_t)res[i] - pred);
        }
    } else if (plan.predictor_type == 1) {
        for (uint32_t i = n; i-- > 2u;) {
            const int64_t pred = (3LL * (int64_t)res[i - 1] - (int64_t)res[i - 2]) >> 2;
            res[i] = (int32_t)((int64_t)res[i] - pred);
        }
    } else {
        for (uint32_t i = n; i-- > 0u;) {
            const int taps = (uint32_t)order < i ? order : (int)i;
            int64_t acc = 0;
            for (int t = 1; t <= taps; ++t) acc += (int64_t)plan.coef[t - 1] * (int64_t)res[i - (uint32_t)t];
            res[i] = (int32_t)((int64_t)res[i] - (acc >> 15));
        }
    }
}

template <class Model>
void emit_partition(BitSink& w, const int32_t* r, uint32_t len, uint32_t mode, uint32_t k0) {
    Model m;
    uint32_t k = k0;
    if (mode == 0) {  // adaptive Rice (ref encoder.cpp:585-600)
        for (uint32_t i = 0; i < len; ++i) {
            const uint32_t u = zigzag32(r[i]);
            w.rice(u, k);
            k = m.push(u);
        }
    } else if (mode == 3) {  // static Rice (ref encoder.cpp:602-607); q forced to 0 at k >= 31 (:79-87)
        for (uint32_t i = 0; i < len; ++i) {
            const uint32_t u = zigzag32(r[i]);
            if (k0 >= 31u) {
                w.put(0, 1);
                w.put(u & ((1u << k0) - 1u), k0);
            } else {
                w.rice(u, k0);
            }
        }
    } else if (mode == 2) {  // bin (ref encoder.cpp:609-667)
        for (uint32_t i = 0; i < len; ++i) {
            const int32_t v = r[i];
            const uint32_t u = zigzag32(v);
            if (v == 0) {
                w.put(0, 2);
            } else if (v == 1 || v == -1) {
                w.put((1u << 1) | (v < 0), 3);
            } else if (v == 2 || v == -2) {
                w.put((2u << 1) | (v < 0), 3);
            } else {
                w.put(3, 2);
                w.rice(u, k);
            }
            k = m.push(u);
        }
    } else {  // zero-run (ref encoder.cpp:669-771)
        uint32_t i = 0;
        while (i < len) {
            uint32_t run = 0;
            while (i + run < len && r[i + run] == 0) ++run;
            if (run >= 4u) {
                w.put(1, 2);
                w.rice(run - 4u, 2);
                for (uint32_t j = 0; j < run; ++j) k = m.push(0);
                i += run;
                continue;
            }
            const uint32_t u = zigzag32(r[i]);
            const uint32_t esc = 1u << ((k + 3u) < 24u ? (k + 3u) : 24u);
            if (u > esc) {
                w.put(2, 2);
                w.put(u, 32);
            } else {
                w.put(0, 2);
                w.rice(u, k);
            }
            k = m.push(u);
            ++i;
        }
    }
}

}  // namespace

size_t emit_channel(const ChannelPlan& plan, const int32_t* a, const int32_t* b, int kind, uint32_t n,
                    uint8_t* out, size_t cap, int32_t* scratch) {
    compute_residual(plan, a, b, kind, n, scratch);
    BitSink w(out, cap);
    w.set_start(out);
    w.put(plan.predictor_type, 8);
    w.put(plan.order, 8);
    if (plan.predictor_type == 2) {
        for (int i = 0; i < plan.order; ++i) w.put((uint16_t)plan.coef[i], 16);
    }
    const uint32_t p = plan.partition_order;
    const uint32_t parts = p ? (1u << p) : 1u;
    uint32_t control = ((uint32_t)(plan.part_mode_k[0] >> 5) & 3u) << 5;  // ref encoder.cpp:773-778
    if (p) control |= 0x80u | (p & 0x0Fu);
    w.put(control, 8);
    for (uint32_t i = 0; i < parts; ++i) w.put(plan.part_mode_k[i] & 0x7Fu, 7);
    const uint32_t base = p ? (n >> p) : n;
    uint32_t off = 0;
    for (uint32_t i = 0; i < parts; ++i) {
        const uint32_t len = (i + 1u == parts) ? (n - off) : base;
        const uint32_t mode = (plan.part_mode_k[i] >> 5) & 3u, k = plan.part_mode_k[i] & 31u;
        if (p) {
            emit_partition<MeanModel>(w, scratch + off, len, mode, k);
        } else {
            emit_partition<DriftModel>(w, scratch + off, len, mode, k);
        }
        off += len;
    }
    return w.finish();
}

uint32_t block_payload_bytes(const StreamParams& sp, const BlockPlan& bp, const ChannelPlan* slots) {
    if (sp.channels == 1) return slots[CH_L].payload_bytes;
    const bool ms = bp.choose_ms != 0;
    const uint32_t pair = ms ? slots[CH_M].payload_bytes + slots[CH_S].payload_bytes
                             : slots[CH_L].payload_bytes + slots[CH_R].payload_bytes;
    return pair + (sp.stereo_mode == 2 ? 1u : 0u);
}

bool emit_one_block(const StreamParams& sp, const int32_t* left, const int32_t* right, uint64_t frames,
                    const BlockPlan& bp, const ChannelPlan* slots, uint32_t b, uint8_t* out, size_t cap,
                    int32_t* scratch) {
    const uint64_t start = (uint64_t)b * kMaxBlock;
    const uint64_t rem = frames - start;
    const uint32_t n = rem < (uint64_t)kMaxBlock ? (uint32_t)rem : (uint32_t)kMaxBlock;
    int kinds[2];
    int nch = 1;
    if (sp.channels == 1) {
        kinds[0] = CH_L;
    } else {
        nch = 2;
        const bool ms = bp.choose_ms != 0;
        kinds[0] = ms ? CH_M : CH_L;
        kinds[1] = ms ? CH_S : CH_R;
        if (sp.stereo_mode == 2) {  // per-block flag byte (ref lac/encoder.cpp:363)
            if (cap == 0) return false;
            *out++ = ms ? 1 : 0;
            --cap;
        }
    }
    for (int c = 0; c < nch; ++c) {
        const ChannelPlan& pl = slots[kinds[c]];
        if (!pl.valid) return false;
        const int32_t* a = left + start;
        const int32_t* bb = right ? right + start : nullptr;
        const size_t wrote = emit_channel(pl, a, bb, kinds[c], n, out, cap, scratch);
        if (wrote == (size_t)-1 || wrote != pl.payload_bytes) return false;
        out += wrote;
        cap -= wrote;
    }
    return cap == 0;
}

struct EmitPool::Impl {
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    bool stop = false;
    uint64_t generation = 0;
    // current job
    StreamParams sp{};
    const int32_t* left = nullptr;
    const int32_t* right = nullptr;
    uint64_t frames = 0;
    const BlockPlan* bplans = nullptr;
    const ChannelPlan* plans = nullptr;
    const uint64_t* offsets = nullptr;
    uint8_t* payload = nullptr;
    uint32_t nblocks = 0;
    std::atomic<uint32_t> next{0}, ready{0}, done{0};
    std::atomic<int> failed{0};
    unsigned active = 0;  // workers inside the current job

    void run_job(std::vector<int32_t>& scratch) {
        for (;;) {
            const uint32_t b = next.fetch_add(1, std::memory_order_relaxed);
            if (b >= nblocks) return;
            // wait until the block is published (or the job is aborted)
            if (ready.load(std::memory_order_acquire) <= b) {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return ready.load(std::memory_order_acquire) > b || failed.load() != 0; });
            }
            if (failed.load() == 0) {
                const size_t cap = (size_t)(offsets[b + 1] - offsets[b]);
                if (!emit_one_block(sp, left, right, frames, bplans[b], plans + (size_t)b * kSlotsPerBlock, b,
                                    payload + offsets[b], cap, scratch.data()))
                    failed.store(1);
            }
            done.fetch_add(1, std::memory_order_release);
        }
    }

    void worker_main() {
        std::vector<int32_t> scratch(kMaxBlock);
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
                ++active;
            }
            run_job(scratch);
            {
                std::lock_guard<std::mutex> lk(mu);
                --active;
            }
            cv_done.notify_all();
        }
    }
};

EmitPool::EmitPool(unsigned threads) : impl_(new Impl) {
    // `threads` extra workers, exactly: 0 means none (finish() runs the job on the calling thread, which is always
    // the last worker).  "Auto" is resolved by the callers (pool_of in api_core.cpp, emit_blocks below).
    impl_->workers.reserve(threads);
    for (unsigned t = 0; t < threads; ++t) impl_->workers.emplace_back([this] { impl_->worker_main(); });
}

EmitPool::~EmitPool() {
    {
        std::lock_guard<std::mutex> lk(impl_->mu);
        impl_->stop = true;
    }
    impl_->cv_work.notify_all();
    for (auto& t : impl_->workers) t.join();
    delete impl_;
}

unsigned EmitPool::threads() const { return (unsigned)impl_->workers.size(); }

void EmitPool::begin(const StreamParams& sp, const int32_t* left, const int32_t* right, uint64_t frames,
                     const BlockPlan* bplans, const ChannelPlan* plans, uint32_t nblocks, const uint64_t* offsets,
                     uint8_t* payload) {
    {
        std::unique_lock<std::mutex> lk(impl_->mu);
        impl_->cv_done.wait(lk, [&] { return impl_->active == 0; });  // no straggler of the previous job
        impl_->sp = sp;
        impl_->left = left;
        impl_->right = right;
        impl_->frames = frames;
        impl_->bplans = bplans;
        impl_->plans = plans;
        impl_->offsets = offsets;
        impl_->payload = payload;
        impl_->nblocks = nblocks;
        impl_->next.store(0);
        impl_->ready.store(0);
        impl_->done.store(0);
        impl_->failed.store(0);
        ++impl_->generation;
    }
    impl_->cv_work.notify_all();
}

void EmitPool::publish(uint32_t ready_upto) {
    {
        std::lock_guard<std::mutex> lk(impl_->mu);
        impl_->ready.store(ready_upto, std::memory_order_release);
    }
    impl_->cv_work.notify_all();
}

void EmitPool::abort() {
    {
        std::lock_guard<std::mutex> lk(impl_->mu);
        impl_->failed.store(1);
    }
    impl_->cv_work.notify_all();
}

bool EmitPool::finish() {
    {
        // the calling thread helps; it also guarantees progress with a pool of size zero
        std::vector<int32_t> scratch(kMaxBlock);
        impl_->run_job(scratch);
    }
    std::unique_lock<std::mutex> lk(impl_->mu);
    impl_->cv_done.wait(lk, [&] { return impl_->done.load(std::memory_order_acquire) >= impl_->nblocks && impl_->active == 0; });
    return impl_->failed.load() == 0;
}

std::string emit_blocks(const StreamParams& sp, const int32_t* left, const int32_t* right, uint64_t frames,
                        const BlockPlan* bplans, const ChannelPlan* plans, uint32_t nblocks,
                        const uint64_t* offsets, uint8_t* payload, uint64_t payload_size, unsigned threads) {
    std::vector<uint64_t> offs(offsets, offsets + nblocks);
    offs.push_back(payload_size);
    unsigned nt = threads ? threads : std::thread::hardware_concurrency();
    if (nt > nblocks) nt = nblocks;
    EmitPool pool(nt > 1 ? nt - 1 : 0);
    pool.begin(sp, left, right, frames, bplans, plans, nblocks, offs.data(), payload);
    pool.publish(nblocks);
    if (!pool.finish()) return "emitted size disagrees with the device plan (internal error)";
    return std::string();
}

void write_frame_header(const StreamParams& sp, uint8_t* o) {
    o[0] = 0x4C;
    o[1] = 0x41;
    o[2] = 3;
    o[3] = sp.channels;
    o[4] = sp.stereo_mode;
    o[5] = (uint8_t)((sp.sample_rate >> 8) & 0xFF);
    o[6] = (uint8_t)(sp.sample_rate & 0xFF);
    o[7] = (uint8_t)((sp.sample_rate >> 16) & 0xFF);
    o[8] = sp.bit_depth;
    o[9] = 0;
}

}  // namespace lacx
