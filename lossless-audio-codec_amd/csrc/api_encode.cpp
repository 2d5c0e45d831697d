// api_encode.cpp -- the encode entry points of the C ABI (include/lacx.h): whole streams, shards, batches, WAV images,
// single blocks (Block::Encoder), container assembly, and the kernel-level probes the parity tests use.
#include "encoder_impl.h"

namespace lacx_host {

// Device buffer for a WAV data chunk of `bytes` bytes (+ the look-ahead the staging loads may touch).
static int ensure_raw(lacx_encoder* e, uint64_t bytes) {
    if (bytes + 16u > e->d_raw_cap) {
        if (e->d_raw) (void)hipFree(e->d_raw);
        e->d_raw = nullptr;
        e->d_raw_cap = 0;
        HIP_TRY(e, hipMalloc((void**)&e->d_raw, bytes + 16u), "hipMalloc(wav data)");
        e->d_raw_cap = bytes + 16u;
    }
    return LACX_OK;
}

// One lane's work in a fan-out, and what a plain encoder does with host input: the frames of hs on e's device, upload
// pipelined with the kernels; results are views into e's buffers.
int encode_host_shard_view(lacx_encoder* e, const HostSrc& hs, int layout, int channels, uint64_t frames, const uint8_t** payload,
                           uint64_t* payload_size, const uint32_t** table, uint32_t* nblocks) {
    int rc = prepare(e, hs.p0, frames);
    if (rc) return rc;
    const int32_t* left = reinterpret_cast<const int32_t*>(hs.p0);
    const int32_t* right = reinterpret_cast<const int32_t*>(hs.p1);
    if (layout == 0 && (e->cfg.flags & LACX_FLAG_HOST_EMIT)) {  // north_star layout: plans back, bit emit on host threads
        rc = upload(e, left, right, frames);
        if (rc) return rc;
        return lacx_encode_shard_device_view(e, e->d_left, right ? e->d_right : nullptr, left, right, frames, nullptr, payload,
                                             payload_size, table, nblocks);
    }
    const int32_t *dl, *dr = nullptr;
    if (layout == 0) {
        rc = ensure_pcm(e, frames, right != nullptr);
        if (rc) return rc;
        dl = e->d_left;
        dr = right ? e->d_right : nullptr;
    } else {
        rc = ensure_raw(e, frames * hs.frame_bytes);
        if (rc) return rc;
        dl = reinterpret_cast<const int32_t*>(e->d_raw);
    }
    uint64_t pay = 0;
    rc = encode_pipelined_device(e, dl, dr, frames, nullptr, &pay, layout, channels, &hs);
    if (rc == -1) {
        if (layout != 0) return fail(e, LACX_E_RUNTIME, "payload exceeds the pinned result reservation");
        // (only with LACX_EMIT_STAGED) the host-emit pipeline; the PCM is on the device already
        return lacx_encode_shard_device_view(e, dl, dr, left, right, frames, nullptr, payload, payload_size, table, nblocks);
    }
    if (rc) return rc;
    *payload = e->h_payload;
    *payload_size = pay;
    *table = e->h_table;
    *nblocks = blocks_for(frames);
    return LACX_OK;
}

}  // namespace lacx_host

extern "C" {

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
    if (is_fanout(e)) {  // the blocks spread over the encoder's devices (api_fanout.cpp)
        HostSrc hs;
        hs.p0 = reinterpret_cast<const uint8_t*>(left);
        hs.p1 = reinterpret_cast<const uint8_t*>(right);
        hs.frame_bytes = sizeof(int32_t);
        return fanout_encode_host(e, hs, 0, right ? 2 : 1, frames, true, out, out_size);
    }
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
    if (is_fanout(e)) {
        HostSrc fs;
        fs.p0 = wav + w.data_offset;
        fs.frame_bytes = (uint64_t)w.channels * (w.bit_depth / 8u);
        uint8_t* buf = nullptr;
        rc = fanout_encode_host(e, fs, w.bit_depth == 16 ? (int)LACX_PCM_INTERLEAVED_I16 : (int)LACX_PCM_INTERLEAVED_I24,
                                (int)w.channels, w.frames, false, &buf, out_size);
        *out = buf;
        return rc;
    }
    rc = ensure_raw(e, w.data_bytes);
    if (rc) return rc;
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

int lacx_debug_stamps(unsigned long long* out32) {
    const int rc = debug_read_stamps(out32);
    if (rc) (void)debug_read_offset_stamps(out32 + 33);  // k_offsets: start, loads back, scan barrier, stores issued (100 MHz ticks)
    return rc;
}

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
