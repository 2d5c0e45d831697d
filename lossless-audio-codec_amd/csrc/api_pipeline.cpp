// api_pipeline.cpp -- the launch pipelines: analysis-only, host-emit pipeline (plans D2H, emit workers), device-emit
// pipeline (fused emit, streaming packer, copy-engine drain, repair kernels, regrow), and the many-streams-as-one-job batch.
// Encode calls are pipelined: the stream is cut into chunks of blocks whose kernels alternate between a few HIP streams.
#include "encoder_impl.h"

namespace lacx_host {

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
    const std::vector<Chunk> chunks = plan_chunks(e->knobs, nb);
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
    x.w.work_ctr = e->d_work_ctr + 8 * c;
    return x;
}

// Size of the pinned result reservation: 1.25 x the PCM at its source bit depth covers every realistic stream (the
// exact size is only known after the analysis; a stream that needs more is re-emitted into a regrown buffer, see
// reemit_into_regrown_buffer).  LACX_PINNED_CAP_BYTES overrides the estimate (tests force the regrow path with it).
uint64_t pinned_reservation(const lacx_encoder* e, uint64_t frames, int channels, uint32_t nb) {
    if (e->knobs.pinned_cap_bytes > 0) return e->knobs.pinned_cap_bytes;
    return frames * (uint64_t)channels * (e->cfg.bit_depth / 8u) * 5u / 4u + (uint64_t)nb * 64u + 4096u;
}

// Copy-engine drain of the payload: every range of stream indices the packer has reported complete (a pinned word per
// range) is fetched from the device payload into the pinned result buffer with hipMemcpyAsync on a copy stream of its
// own -- a copy engine, not CUs.  Called from wherever the calling thread waits: the launch phase of an encode whose
// input is still uploading (chunk c's payload leaves while chunk c + 1's PCM arrives: PCIe is full duplex) and the wait
// for the kernels in encode_device_end.
void drain_pump(lacx_encoder* e) {
    if (!e->pend.drained || e->pend.ranges == 0) return;
    const volatile unsigned long long* flags = e->h_range;
    while (e->pend.next_range < e->pend.ranges) {
        const unsigned long long v = flags[e->pend.next_range];
        if (v == 0) break;
        const uint64_t end = v - 1u;
        if (e->knobs.debug_drain)
            std::fprintf(stderr, "[drain] range %u end %llu at %.3f ms\n", e->pend.next_range, (unsigned long long)end, ms_since(e->pend.t0));
        if (end > e->pend.drained_to && end <= e->h_payload_cap) {
            hipStream_t cs = (e->pend.next_range & 1u) && e->knobs.two_copy_streams ? e->copy_stream2 : e->copy_stream;
            if (hipMemcpyAsync(e->h_payload + e->pend.drained_to, e->d_payload + e->pend.drained_to, end - e->pend.drained_to,
                               hipMemcpyDeviceToHost, cs) != hipSuccess)
                return;  // (the final copy in encode_device_end fetches what is missing)
            e->pend.drained_to = end;
            const double now = ms_since(e->pend.t0);
            if (e->timing.drain_copies == 0) e->timing.drain_first_ms = now;
            e->timing.drain_last_ms = now;
            e->timing.drain_copies += 1;
        }
        ++e->pend.next_range;
    }
}

// Part 1: enqueue everything (no host synchronisation).
int encode_device_begin_impl(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, uint64_t frames,
                             hipStream_t user_stream, int layout, int layout_channels, const HostSrc* hs);
int encode_device_begin(lacx_encoder* e, const int32_t* d_left, const int32_t* d_right, uint64_t frames,
                        hipStream_t user_stream, int layout, int layout_channels, const HostSrc* hs) {
    const int rc = encode_device_begin_impl(e, d_left, d_right, frames, user_stream, layout, layout_channels, hs);
    // A failure half-way leaves kernels queued that write to the workspace, the slots and the pinned result buffer: they
    // must have drained before the next call clears, frees or regrows any of those.
    if (rc != LACX_OK && e->uploader) e->uploader->wait();  // (it reads the caller's buffer and writes this encoder's)
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
    const Knobs& kn = e->knobs;
    const bool fused = kn.fused_emit;
    e->pend.chunks = plan_chunks(kn, nb, true, fused, hs != nullptr);
    const std::vector<Chunk>& chunks = e->pend.chunks;
    // Destination of k_emit: by default the pinned host buffer itself (the kernel's 16-byte stores cross PCIe
    // while later blocks are still being analysed, so no separate D2H pass is left at the end); with
    // LACX_EMIT_STAGED=1 a device arena sized for the worst case (12 bytes per sample), copied afterwards.
    const bool staged = kn.emit_staged;
    // Default with the fused emit: the packer packs into device memory and a copy engine drains it (LACX_DIRECT_PACKER=1:
    // the packer's CUs store straight into pinned host memory, the round-2 layout).
    const bool drained = !staged && fused && !kn.direct_packer && kn.packer && kn.pinned_cap_bytes == 0;
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
    if (host_cap > e->h_payload_cap || prefix > e->h_prefix || kn.pinned_cap_bytes) {
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
    e->timing.drain_copies = 0;
    e->timing.drain_first_ms = e->timing.drain_last_ms = e->timing.poll_gap_max_ms = e->timing.kernels_done_ms = 0;
    const auto t0 = clk::now();
    e->pend.t0 = t0;
    e->pend.drained = false;  // (set once the packer is launched: drain_pump looks at it)
    e->pend.ranges = 0;
    e->pend.next_range = 0;
    e->pend.drained_to = 0;
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
    if (hs) {
        if (!e->uploader) {
            e->uploader.reset(new Uploader());
            HIP_TRY(e, hipStreamCreateWithFlags(&e->up_stream, hipStreamNonBlocking), "hipStreamCreate");
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            HIP_TRY(e, hipStreamCreateWithPriority(&e->front_stream, hipStreamNonBlocking, greatest), "hipStreamCreate");
            for (auto& ev : e->up_ev) HIP_TRY(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
        }
        for (auto& d : e->up_done) d.store(0, std::memory_order_relaxed);
        e->up_ms = 0;
        const HostSrc src = *hs;
        const std::vector<Chunk> plan = chunks;
        const int dev = e->device;
        uint8_t* dst0 = const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(d_left));
        uint8_t* dst1 = const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(d_right));
        e->uploader->post([e, src, plan, dev, dst0, dst1, frames] {
            (void)hipSetDevice(dev);
            const auto tu0 = clk::now();
            bool ok = true;
            for (size_t c = 0; c < plan.size(); ++c) {
                const uint64_t f0 = (uint64_t)plan[c].first * kMaxBlock;
                const uint64_t f1 = std::min<uint64_t>(frames, (uint64_t)(plan[c].first + plan[c].count) * kMaxBlock);
                const uint64_t o = f0 * src.frame_bytes, nbytes = (f1 - f0) * src.frame_bytes;
                if (ok) ok = hipMemcpyAsync(dst0 + o, src.p0 + o, nbytes, hipMemcpyHostToDevice, e->up_stream) == hipSuccess;
                if (ok && src.p1) ok = hipMemcpyAsync(dst1 + o, src.p1 + o, nbytes, hipMemcpyHostToDevice, e->up_stream) == hipSuccess;
                if (ok) ok = hipEventRecord(e->up_ev[c], e->up_stream) == hipSuccess;
                e->up_done[c].store(ok ? 1 : -1, std::memory_order_release);
            }
            e->up_ms = ms_since(tu0);
        });
    }
    // Upload pipeline with the front kernels on their own (highest priority) stream: the chunks' whole-block kernels then
    // run on the encoder's lower-priority streams only (stream 0 shares the top priority with the front stream).
    const bool front_split = hs && kn.front_stream_split && chunks.size() > 1 && !user_stream;
    auto chunk_stream = [&](size_t c) { return front_split ? st[1 + c % (kStreams - 1)] : st[c % kStreams]; };
    for (size_t c = 0; c < chunks.size(); ++c) {
        const Chunk& ck = chunks[c];
        hipStream_t s = chunk_stream(c);
        if (s != st[0]) HIP_TRY(e, hipStreamWaitEvent(s, e->prologue, 0), "stream wait");
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
            fa.silent = e->knobs.silent_template ? e->d_silent : nullptr;
            fa.silent_copies = e->ws.err_flag + kMaxChunks + 4;
        }
        if (hs) {
            // this chunk's PCM is being copied by the uploader thread: wait (on the host) until its copy has been issued and
            // its event recorded, then make the chunk's stream wait for that event
            const auto tw0 = clk::now();
            int st_up = 0;
            while ((st_up = e->up_done[c].load(std::memory_order_acquire)) == 0) {
                drain_pump(e);  // earlier chunks' payload leaves while this chunk's PCM arrives
                __builtin_ia32_pause();
                if (ms_since(tw0) > 20000.0) break;
            }
            if (st_up != 1) {
                e->uploader->wait();
                return fail(e, LACX_E_DEVICE, "host to device copy of the PCM failed");
            }
            HIP_TRY(e, hipStreamWaitEvent(s, e->up_ev[c], 0), "stream wait");
            if (front_split) {
                HIP_TRY(e, hipStreamWaitEvent(e->front_stream, e->up_ev[c], 0), "stream wait");
                if (c == 0) HIP_TRY(e, hipStreamWaitEvent(e->front_stream, e->prologue, 0), "stream wait");
            }
        }
        // Fused emit: the packer walks the stream indices in order, so the whole-block kernels of the chunks run in that
        // order too (chunk c's waits for chunk c-1's: ev[c-1][3] is recorded behind it); what comes before them --
        // ingest, Levinson, probes -- still overlaps the previous chunk's analysis.
        const bool chain = fused && c > 0 && kn.chain;
        (void)cl;
        (void)cr;
        (void)prm;
        LaunchSet ls = cx.set(fuse_items, emit_cap);
        // persistent analysis workgroups only for a shard that is one chunk: with several, the next chunk's ingest /
        // Levinson / probe kernels are meant to run beside this chunk's analysis, which persistent workgroups would not let in
        DeviceWorkspace wl = w;
        if (chunks.size() > 1 || !kn.persistent) wl.work_ctr = nullptr;
        LaunchTuning tune = kn.tune;
        if (front_split) tune.front_stream = e->front_stream;
        if (kn.front_halves && chunks.size() == 1 && s != e->stream[kStreams - 1]) {  // (see LaunchTuning::aux_stream)
            tune.aux_stream = e->stream[kStreams - 1];
            tune.aux_ev[0] = e->aux_ev[0];
            tune.aux_ev[1] = e->aux_ev[1];
        }
        HIP_TRY(e, launch_analysis(bind(ls), wl, s, e->ev[c], &fa, chain ? e->ev[c - 1][3] : nullptr, tune), "kernel launch");
        if (c == 0 && fuse_items && kn.packer) {
            // the streaming packer: beside the whole-block analysis kernels, on its own stream.  It starts when the first
            // chunk's ingest / Levinson / probe kernels are done (ev[0][2] is recorded right in front of the whole-block
            // kernel), so its bounded waits only ever cover the progress of the analysis itself, however long the shard.
            // (With persistent analysis workgroups it has to be resident before they are: it then starts in front of the
            // ingest kernel -- ev[0][0], behind the call's memset -- and holds its CUs through the front kernels.)
            HIP_TRY(e, hipStreamWaitEvent(e->pack_stream, e->ev[0][analysis_is_persistent(wl) ? 0 : 2], 0), "stream wait");
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
                rp.fence_mode = kn.drain_fence;
                e->pend.ranges = ranges;
                e->pend.drained = true;
            }
            HIP_TRY(e, launch_stream_out(bind(shard), e->ws, emit_dst, e->ws.err_flag + kMaxChunks, e->pack_stream, rp, kn.tune), "packer launch");
            HIP_TRY(e, hipEventRecord(e->pack_done, e->pack_stream), "event record");
        }
    }
    // Second pass over the chunks: everything behind the analysis.  With the fused emit it waits for the packer (what
    // k_pack / k_emit still have to move is only known once the packer has finished), and a stream may carry several
    // chunks, so none of this may be enqueued before the last chunk's analysis kernels.
    for (size_t c = 0; c < chunks.size(); ++c) {
        const Chunk& ck = chunks[c];
        hipStream_t s = chunk_stream(c);
        const ChunkCtx cx = chunk_ctx(e, d_left, d_right, frames, layout, channels, ck, c);
        const AnalyzeParams& prm = cx.prm;
        const int32_t *cl = cx.left, *cr = cx.right;
        const DeviceWorkspace& w = cx.w;
        // block offsets are global: chunk c starts where chunk c-1 ended (its k_offsets must have run)
        const bool packer_counts = fuse_items && kn.packer;
        // Lazy repair: a one-chunk shard whose channel blocks all take part in the fused emit normally leaves k_pack and
        // k_emit nothing to do; they are not even enqueued, the gather kernel checks the packer's count and the host
        // enqueues them afterwards in the rare case (a packer wave that gave up, a bitstream longer than its slot).
        const bool lazy = kn.lazy_repair && packer_counts && chunks.size() == 1 && fuse_items == nb * (uint32_t)channels;
        (void)cl;
        (void)cr;
        (void)prm;
        LaunchSet ls = cx.set(fuse_items, emit_cap);
        if (lazy) {
            // Not even k_offsets: the block table follows from the size records the analysis kernel published (the host
            // adds them up), the repair kernels -- the only readers of the device-side offsets -- are enqueued on demand,
            // k_offsets in front of them.  All that stands between the end of the analysis and the host is the packer's
            // completion and one gather kernel.
            if (nb * (uint32_t)channels > e->h_sizes_cap) {
                if (e->h_sizes) (void)hipHostFree(e->h_sizes);
                e->h_sizes = nullptr;
                e->h_sizes_cap = 0;
                HIP_TRY(e, hipHostMalloc((void**)&e->h_sizes, (size_t)nb * channels * sizeof(unsigned long long), 0), "hipHostMalloc(sizes)");
                e->h_sizes_cap = nb * (uint32_t)channels;
            }
            HIP_TRY(e, hipStreamWaitEvent(s, e->pack_done, 0), "stream wait");
        } else {
            HIP_TRY(e, launch_emit(bind(ls), w, emit_dst, prev_end, c ? e->copied[c - 1] : nullptr,
                                   e->copied[c], s, true, packer_counts ? e->ws.err_flag + kMaxChunks + 1 : nullptr,
                                   nb * (uint32_t)channels, packer_counts ? e->pack_done : nullptr,
                                   e->ws.err_flag + kMaxChunks + 3, false), "emit launch");
        }
        e->pend.lazy_repair = lazy;
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
        if (lazy) {
            unsigned long long* m_sizes = mapped(e->h_sizes);
            if (!m_sizes) return fail(e, LACX_E_RUNTIME, "hipHostGetDevicePointer failed");
            g.add(e->ws.size_rec, m_sizes, (size_t)nb * channels * sizeof(unsigned long long));
            // the "all kernels done" word: any record (its valid bit makes it non-zero); the host puts the total there
            g.add(e->ws.size_rec, &m_totals[c], sizeof(unsigned long long));
        } else {
            g.add(w.table, m_table + (size_t)ck.first * 2, (size_t)ck.count * 2 * sizeof(uint32_t));
            g.add(w.block_off + ck.count, &m_totals[c], sizeof(unsigned long long));
        }
        g.add(w.err_flag, &m_err[c], sizeof(uint32_t));
        // the packer's error flags, moved count, waves that gave up, and k_pack's repacked count
        if (c + 1 == chunks.size()) g.add(e->ws.err_flag + kMaxChunks, &m_err[kMaxChunks], 5 * sizeof(uint32_t));
        if (fused)
            g.add(e->ws.packed + (size_t)ck.first * channels, m_emitted + (size_t)ck.first * channels,
                  (size_t)ck.count * channels * sizeof(uint32_t));
        g.add(w.t_first, &m_tspan[c], sizeof(unsigned long long));
        g.add(w.t_last, &m_tspan[kMaxChunks + c], sizeof(unsigned long long));
        HIP_TRY(e, launch_gather(g, s), "gather launch");
        HIP_TRY(e, hipEventRecord(e->done[c], s), "event record");
    }
    e->timing.enqueue_ms = ms_since(t0);
    e->pend.active = true;
    e->pend.nb = nb;
    e->pend.channels = channels;
    e->pend.staged = staged;
    e->pend.fused = fused;
    e->pend.drained = drained;  // (the result is fetched from the device payload even when no range was ever reported)
    if (!(drained && fused && fuse_items != 0)) e->pend.ranges = 0;
    for (int i = 0; i < kStreams; ++i) e->pend.st[i] = st[i];
    e->pend.front_split = front_split;
    e->pend.fuse_items = fuse_items;
    e->pend.emit_cap = emit_cap;
    e->pend.emit_dst = emit_dst;
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
    if (e->uploader) {  // (long finished: every chunk's kernels were enqueued behind its copy)
        e->uploader->wait();
        if (e->up_ms > 0) e->timing.h2d_ms = e->up_ms;
        e->up_ms = 0;
    }
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
    if (e->pend.drained) {
        auto pump = [&]() { drain_pump(e); };
        // (no runtime call in the loop but the copies: the gather kernel -- the last one of the call -- stores the
        // cumulative byte count of the last chunk, non-zero, into pinned memory that was zeroed before the launch)
        const volatile unsigned long long* finished = &e->h_totals[chunks.size() - 1];
        // The loop is a spin on pinned memory with a pause instruction between looks (a sibling hyper-thread keeps its
        // issue slots); every look is time-stamped, so a host thread that was descheduled or busy elsewhere shows up as
        // poll_gap_max_ms instead of as an unexplained long step.  Once a millisecond the last chunk's event is queried as
        // well: a failed device ends the wait even though its completion word never arrives.
        const auto poll0 = clk::now();
        auto last_look = poll0;
        auto last_query = poll0;
        while (*finished == 0ull) {
            pump();
            const auto now = clk::now();
            const double gap = std::chrono::duration<double, std::milli>(now - last_look).count();
            if (gap > e->timing.poll_gap_max_ms) e->timing.poll_gap_max_ms = gap;
            last_look = now;
            if (std::chrono::duration<double, std::milli>(now - last_query).count() > 1.0) {
                last_query = now;
                const hipError_t qe = hipEventQuery(e->done[chunks.size() - 1]);
                if (qe != hipErrorNotReady) break;  // done (the word is about to follow) or failed: the event wait below reports it
                if (std::chrono::duration<double, std::milli>(now - poll0).count() > 20000.0) break;
            }
            __builtin_ia32_pause();
        }
        e->timing.kernels_done_ms = ms_since(t0);
        if (e->knobs.debug_drain) std::fprintf(stderr, "[drain] kernels done at %.3f ms, copy stream %s\n", ms_since(t0),
                                               hipStreamQuery(e->copy_stream) == hipSuccess ? "idle" : "busy");
        pump();
    }
    uint64_t drained_to = e->pend.drained_to;
    if (e->pend.lazy_repair && hipEventSynchronize(e->done[0]) == hipSuccess) {
        // the block table and the payload's size from the size records (k_offsets never ran on this path)
        unsigned long long total = 0;
        bool complete = true;
        for (uint32_t b = 0; b < nb; ++b) {
            unsigned long long bytes = 0;
            for (int ch = 0; ch < channels; ++ch) {
                const unsigned long long rec = e->h_sizes[(size_t)b * channels + ch];
                complete = complete && (rec >> 62) == 1ull;  // (kRecValid)
                bytes += rec & ((1ull << 60) - 1ull);
            }
            e->h_table[2 * (size_t)b] = e->h_bplans[b].frames;
            e->h_table[2 * (size_t)b + 1] = (uint32_t)bytes;
            total += bytes;
        }
        e->h_totals[0] = total;
        bool any_invalid = false;
        for (uint32_t b = 0; b < nb; ++b) any_invalid = any_invalid || e->h_bplans[b].invalid;
        if (!complete && !any_invalid) status = fail(e, LACX_E_RUNTIME, "a channel block's size record is missing (internal error)");
    }
    if (status == LACX_OK && e->pend.lazy_repair && e->h_err[kMaxChunks + 1] != nb * (uint32_t)channels) {
        // Lazy repair (see launch_emit): the packer did not move every channel block -- a wave gave up, a bitstream did
        // not fit its slot.  Now the repair kernels run: k_pack for the slots left behind, k_emit for what was never
        // emitted, then the gather once more.  (One chunk; its k_offsets has run, the offsets are in place.)
        const ChunkCtx cx = chunk_ctx(e, e->pend.d_left, e->pend.d_right, e->pend.frames, e->pend.layout, channels, chunks[0], 0);
        LaunchSet ls = cx.set(e->pend.fuse_items, e->pend.emit_cap);
        hipStream_t s = st[0];
        hipError_t re = launch_emit(bind(ls), cx.w, e->pend.emit_dst, nullptr, nullptr, nullptr, s, true, e->ws.err_flag + kMaxChunks + 1,
                                    nb * (uint32_t)channels, nullptr, e->ws.err_flag + kMaxChunks + 3, false);
        GatherList g;
        auto mapped = [](auto* host) -> decltype(host) {
            void* d = nullptr;
            return hipHostGetDevicePointer(&d, host, 0) == hipSuccess ? static_cast<decltype(host)>(d) : nullptr;
        };
        uint32_t *m_err = mapped(e->h_err), *m_emitted = mapped(e->h_emitted);
        uint32_t* m_table = mapped(e->h_table);
        unsigned long long* m_totals = mapped(e->h_totals);
        if (re == hipSuccess && m_err && m_emitted && m_table && m_totals) {
            g.add(cx.w.table, m_table, (size_t)nb * 2 * sizeof(uint32_t));
            g.add(cx.w.block_off + nb, &m_totals[0], sizeof(unsigned long long));
            g.add(cx.w.err_flag, &m_err[0], sizeof(uint32_t));
            g.add(e->ws.err_flag + kMaxChunks, &m_err[kMaxChunks], 5 * sizeof(uint32_t));
            g.add(e->ws.packed, m_emitted, (size_t)nb * channels * sizeof(uint32_t));
            re = launch_gather(g, s);
        }
        if (re == hipSuccess) re = hipEventRecord(e->done[0], s);
        if (re != hipSuccess) status = hip_fail(e, re, "repair launch");
        e->timing.regrows += 0;  // (not a regrow; the counters below say what happened: moved_by_k_pack, packer_gave_up)
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
            hipStream_t s = e->pend.front_split ? st[1 + c % (kStreams - 1)] : st[c % kStreams];
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
            if (e->h_err[kMaxChunks] & 4u) {
                // (range copies of the stale bytes may still be queued on either copy stream: they must have landed before
                // the full copy is enqueued, or one of them could overwrite what k_emit / k_pack placed later)
                (void)hipStreamSynchronize(e->copy_stream2);
                (void)hipStreamSynchronize(e->copy_stream);
                drained_to = 0;
            }
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
    e->timing.emit_direct = e->timing.moved_by_k_pack = e->timing.packer_gave_up = e->timing.silent_copies = 0;
    if (e->pend.fused) {
        for (size_t i = 0; i < (size_t)nb * (size_t)channels; ++i) e->timing.emit_direct += e->h_emitted[i] == 1u;
        e->timing.packer_gave_up = e->h_err[kMaxChunks + 2];
        e->timing.moved_by_k_pack = e->h_err[kMaxChunks + 3];
        e->timing.silent_copies = e->h_err[kMaxChunks + 4];
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
                            hipStream_t user_stream, uint64_t* payload_size, int layout, int layout_channels,
                            const HostSrc* hs) {
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

// ---- many streams as ONE launch set (lacx_encode_batch_device) ------------------------------------------------------
// Every stream keeps its own parameters (rate, depth, channels, stereo mode, layout); the kernels resolve the stream of a
// block from the descriptor table (StreamDesc, lacx_types.h).  One ingest / Levinson / probe / whole-block launch over
// all blocks of all streams, one packer; every stream's payload lands in its own region of the pinned result buffer.
int encode_batch(lacx_encoder* e, const lacx_batch_item* items, uint32_t n, hipStream_t user_stream, lacx_batch_out* out,
                 const std::vector<uint64_t>* exact_caps) {
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
        if (e->knobs.pinned_cap_bytes) sd.out_cap = e->knobs.pinned_cap_bytes;  // (tests force the second attempt with it)
        if (exact_caps) sd.out_cap = (*exact_caps)[i];
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
    e->h_payload = e->h_payload_base + e->h_prefix;  // (the regions start where the shard path's payload does)
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
    w.work_ctr = e->knobs.persistent ? e->d_work_ctr : nullptr;
    FuseArgs fa;
    fa.slots = e->ws.slots;
    fa.slot_stride = e->ws.slot_stride;
    fa.emitted = e->ws.emitted;
    fa.err_flag = w.err_flag;
    fa.size_rec = e->ws.size_rec;
    fa.ready_rec = e->ws.ready_rec;
    fa.silent = e->knobs.silent_template ? e->d_silent : nullptr;
    fa.silent_copies = e->ws.err_flag + kMaxChunks + 4;
    auto run = [&]() -> int {
        HIP_TRY(e, launch_analysis(ls, w, s, e->ev[0], &fa, nullptr, e->knobs.tune), "kernel launch");
        const bool packer = e->knobs.packer;
        if (packer) {
            HIP_TRY(e, hipStreamWaitEvent(e->pack_stream, e->ev[0][analysis_is_persistent(w) ? 0 : 2], 0), "stream wait");
            HIP_TRY(e, launch_stream_out(ls, e->ws, emit_dst, e->ws.err_flag + kMaxChunks, e->pack_stream, RangeProgress{}, e->knobs.tune), "packer launch");
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
        g.add(e->ws.err_flag + kMaxChunks, &m_err[kMaxChunks], 5 * sizeof(uint32_t));
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
    if ((e->h_err[0] & 2u) || (e->h_err[kMaxChunks] & 2u)) {
        // A stream needs more than its estimated reservation (the single-shard path re-emits into a regrown buffer, the
        // reference never fails on size): k_offsets has run, so the block table holds every stream's exact size -- run the
        // job once more with exact regions.
        if (exact_caps) return fail(e, LACX_E_RUNTIME, "a stream's payload exceeds its exact result reservation (internal error)");
        std::vector<uint64_t> caps(n);
        for (uint32_t i = 0; i < n; ++i) {
            uint64_t bytes = 0;
            for (uint32_t b = 0; b < sds[i].prm.num_blocks; ++b) bytes += e->h_table[2 * ((size_t)sds[i].first_block + b) + 1];
            caps[i] = bytes + 4096u;
        }
        const int rr = encode_batch(e, items, n, user_stream, out, &caps);
        e->timing.regrows += 1;
        return rr;
    }
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
    e->timing.silent_copies = e->h_err[kMaxChunks + 4];
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

}  // namespace lacx_host

