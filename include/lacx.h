/* lacx.h -- C ABI of the MI355X-native LAC block-encode path (liblacx.so).
 *
 * The reference (audexdev/Lossless-Audio-Codec, C++20) has no FFI layer: its encode boundary is two
 * C++ classes.  Each entry point below names the reference interface it replaces; the C++ mirror
 * classes with the reference's own signatures (lossless-audio-codec_amd/include/codec/...) are thin
 * wrappers over this ABI, and INTEGRATION.md shows the binding a reference maintainer would add.
 *
 *   lacx_encode            <- LAC::Encoder::encode          ref src/codec/lac/encoder.hpp:22-24, encoder.cpp:215-466
 *   lacx_encoder_create    <- LAC::Encoder::Encoder + set_zero_run_enabled / set_partitioning_enabled /
 *                             set_thread_count               ref src/codec/lac/encoder.hpp:14-29
 *   lacx_block_encode      <- Block::Encoder::encode        ref src/codec/block/encoder.hpp:15, encoder.cpp:313-838
 *   lacx_analyze           <- the decisions inside Block::Encoder::encode / estimate_stereo_mode
 *                                                            ref block/encoder.cpp:313-552, lac/encoder.cpp:126-197,321-373
 *   lacx_emit_from_plans   <- the emit half of Block::Encoder::encode + container write
 *                                                            ref block/encoder.cpp:554-838, lac/encoder.cpp:243-250,445-465
 *   lacx_wav_parse /
 *   lacx_encode_wav        <- read_wav + LAC::Encoder::encode as chained by the CLI
 *                                                            ref src/io/wav_io.cpp:167-277, src/main.cpp:640-675
 *   lacx_encode_shard /
 *   lacx_assemble          <- the block loop + block table concat of LAC::Encoder::encode, split so that
 *                             contiguous block ranges can be encoded by different GPUs/processes
 *                                                            ref lac/encoder.cpp:252-263, 445-465
 *   lacx_encoder_create_multi /
 *   lacx_encode_fanout_resident <- the worker pool of LAC::Encoder::encode with devices as the workers
 *                                                            ref lac/encoder.cpp:385-443, 445-465
 *   lacx_encode_batch_device <- one LAC::Encoder::encode per file of a corpus, as one device job
 *                                                            ref lac/encoder.cpp:215-466 (block pool :404-435)
 *   lacx_stream_parse /
 *   lacx_decode            <- LAC::Decoder::decode          ref src/codec/lac/decoder.hpp:10-24, decoder.cpp:76-303,
 *                                                            src/codec/block/decoder.cpp:64-520
 *
 * All analysis (and the decode) runs in hand-written HIP kernels on a gfx950 device; there is no CPU fallback: every
 * call that needs the device fails with LACX_E_DEVICE when none is usable.
 */
#ifndef LACX_H
#define LACX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LACX_OK 0
#define LACX_E_INVALID 1 /* maps to std::invalid_argument (ref lac/encoder.cpp:220-241) */
#define LACX_E_RUNTIME 2 /* maps to std::runtime_error   (ref lac/encoder.cpp:447-449) */
#define LACX_E_DEVICE 3  /* HIP failure / no device: std::runtime_error in the C++ mirror */

#define LACX_MAX_BLOCK 16384u
#define LACX_SLOTS_PER_BLOCK 16u /* slot = window*4 + channel(L,R,M,S); window 0 = whole block, 1..3 = probes */

typedef struct lacx_encoder lacx_encoder;

typedef struct lacx_config {
    uint32_t sample_rate;         /* 44100 / 48000 / 96000 / 192000 */
    uint8_t bit_depth;            /* 16 / 24 */
    uint8_t stereo_mode;          /* 0 LR, 1 MS, 2 per-block auto (ignored for mono input) */
    uint8_t zero_run_enabled;     /* reference default: 1 */
    uint8_t partitioning_enabled; /* reference default: 1 */
    int32_t device;               /* HIP device ordinal, -1 = current device, LACX_DEVICE_ALL = every visible device
                                     (whole-stream calls fan the blocks out over them, see lacx_encoder_create_multi) */
    uint32_t emit_threads;        /* host emit worker threads, 0 = hardware concurrency */
    uint32_t flags;               /* LACX_FLAG_* */
} lacx_config;

#define LACX_DEVICE_ALL (-2)
#define LACX_FLAG_HOST_EMIT 1u /* keep the bit emit on the host (north_star layout); default: device-side emit */

/* Same layout as lacx::ChannelPlan (csrc/lacx_types.h). */
typedef struct lacx_channel_plan {
    uint8_t predictor_type; /* 0 fixed, 1 FIR, 2 LPC */
    uint8_t order;
    uint8_t partition_order;
    uint8_t valid;
    int16_t coef[12];
    uint32_t payload_bytes;
    uint64_t total_bits;
    uint8_t part_mode_k[256]; /* (mode << 5) | k */
} lacx_channel_plan;

/* Same layout as lacx::BlockPlan. */
typedef struct lacx_block_plan {
    uint8_t choose_ms;
    uint8_t uncertain;
    uint8_t est_ms;
    uint8_t invalid;
    uint32_t frames;
    uint32_t first_bad;
    uint32_t pad;
} lacx_block_plan;

typedef struct lacx_timing {
    double h2d_ms;          /* host -> device PCM copy (0 for device-resident input) */
    double analysis_ms;     /* all kernels, device timeline (hipEvent) */
    double ingest_ms;       /* k_ingest + k_levinson */
    double probe_ms;        /* k_analyze<4,64> + k_decide */
    double full_ms;         /* k_analyze<16,1024> (the dominant kernel), summed over its launches */
    double d2h_ms;          /* plan records device -> host, incl. stream sync */
    double emit_ms;         /* host emit tail after the last plan arrived, or (device emit) the k_emit kernels */
    double total_ms;        /* wall time of the call */
    uint64_t full_slots;    /* workgroups of the dominant kernel that did work */
    uint64_t probe_slots;
    uint32_t full_launches; /* launches of the dominant kernel in the call (one per pipeline chunk) */
    uint32_t regrows;       /* device emit: times the pinned result buffer had to be regrown and the emit re-run */
    double full_exec_ms;    /* device emit pipeline: k_analyze<16,1024> execution spans (first workgroup start to last
                               workgroup end, device clock), summed over its launches -- full_ms minus queueing */
    uint32_t emit_direct;   /* fused emit: channel blocks the streaming packer moved to the payload beside the analysis */
    uint32_t moved_by_k_pack; /* ... and those the repair kernel k_pack had to move afterwards (0 when the packer kept up) */
    uint32_t packer_gave_up;  /* packer waves that stopped after 20 ms without an awaited record (0 normally; when the
                                 packer cannot run beside the analysis -- a profiler that serialises kernels, a shared
                                 GPU -- every wave gives up and k_pack moves everything: correct, but slower) */
    uint32_t drain_copies;    /* copy-engine drain: range copies issued while the kernels ran (the tail copy not counted) */
    /* The payload drain depends on the calling thread: it polls pinned progress words and issues a copy per completed
       range.  These say how attentive it was (all in ms since the call began; 0 when the drain is not in use): */
    double drain_first_ms;    /* first range copy issued */
    double drain_last_ms;     /* last range copy issued */
    double poll_gap_max_ms;   /* longest interval between two looks at the progress words (a descheduled or busy host thread) */
    double kernels_done_ms;   /* the host saw the last kernel's completion word */
    double enqueue_ms;        /* everything enqueued (the call's launch phase) */
    uint32_t silent_copies;   /* channel blocks of nothing but zeros that were copies of the call's first one (plan and
                                 bitstream are the same for every such block of the same length) */
    uint32_t reserved0;
} lacx_timing;

int lacx_encoder_create(const lacx_config* cfg, lacx_encoder** out);
void lacx_encoder_destroy(lacx_encoder* enc);
const char* lacx_last_error(const lacx_encoder* enc);
void lacx_free(void* p);
void lacx_get_timing(const lacx_encoder* enc, lacx_timing* out);

/* sizeof() of a public struct as this library was built, by name without the prefix ("config", "channel_plan",
 * "block_plan", "timing", "pcm", "batch_item", "batch_out", "wav_info", "fanout_shard", "fanout_out", "fanout_stats",
 * "stream_info"); 0 for an unknown name.  A binding that declares the structs itself (ctypes, cgo, JNI) checks its layout
 * against this before the first call that fills one. */
uint32_t lacx_sizeof(const char* struct_name);

/* Whole-stream encode of host planar int32 PCM (right == NULL => mono). *out is malloc'd; free with
 * lacx_free. Byte-identical to the reference's LAC::Encoder::encode output. */
int lacx_encode(lacx_encoder* enc, const int32_t* left, const int32_t* right, uint64_t frames,
                uint8_t** out, uint64_t* out_size);

/* Same, with the PCM already resident in device memory (d_*), e.g. torch tensors.  h_left/h_right are
 * the host copies the host-side emit reads; if NULL the library copies the PCM back itself.
 * `stream` is a hipStream_t (NULL = the encoder's own stream). */
int lacx_encode_device(lacx_encoder* enc, const int32_t* d_left, const int32_t* d_right,
                       const int32_t* h_left, const int32_t* h_right, uint64_t frames, void* stream,
                       uint8_t** out, uint64_t* out_size);

/* Device analysis only: fills bplans[nblocks] and plans[nblocks * LACX_SLOTS_PER_BLOCK]
 * (nblocks = ceil(frames / 16384)). Host pointers in. */
int lacx_analyze(lacx_encoder* enc, const int32_t* left, const int32_t* right, uint64_t frames,
                 lacx_block_plan* bplans, lacx_channel_plan* plans);
int lacx_analyze_device(lacx_encoder* enc, const int32_t* d_left, const int32_t* d_right, uint64_t frames,
                        void* stream, lacx_block_plan* bplans, lacx_channel_plan* plans);

/* Host-only: emit + container from plans (no device needed). */
int lacx_emit_from_plans(lacx_encoder* enc, const int32_t* left, const int32_t* right, uint64_t frames,
                         const lacx_block_plan* bplans, const lacx_channel_plan* plans, uint8_t** out,
                         uint64_t* out_size);

/* Shard interface for multi-GPU block-range splits: encodes the blocks of a frame range that starts
 * on a block boundary.  Returns the concatenated block payloads and a table of (frames, bytes) pairs
 * (2 * nblocks uint32).  Both malloc'd. */
int lacx_encode_shard(lacx_encoder* enc, const int32_t* left, const int32_t* right, uint64_t frames,
                      uint8_t** payload, uint64_t* payload_size, uint32_t** table, uint32_t* nblocks);
int lacx_encode_shard_device(lacx_encoder* enc, const int32_t* d_left, const int32_t* d_right,
                             const int32_t* h_left, const int32_t* h_right, uint64_t frames, void* stream,
                             uint8_t** payload, uint64_t* payload_size, uint32_t** table,
                             uint32_t* nblocks);

/* Zero-copy variant: *payload / *table point into buffers owned by the encoder (pinned host memory) that
 * stay valid until the next call on the same encoder. */
int lacx_encode_shard_device_view(lacx_encoder* enc, const int32_t* d_left, const int32_t* d_right,
                                  const int32_t* h_left, const int32_t* h_right, uint64_t frames, void* stream,
                                  const uint8_t** payload, uint64_t* payload_size, const uint32_t** table,
                                  uint32_t* nblocks);

/* Device-resident PCM in its source layout: the kernels read the WAV data chunk directly
 * with coalesced loads, 2 or 3 bytes per sample instead of the 4 of the planar int32 API. */
#define LACX_PCM_PLANAR_I32 0u      /* data0 = left, data1 = right (NULL for mono) */
#define LACX_PCM_INTERLEAVED_I16 1u /* data0 = interleaved little-endian int16 frames, 4-byte aligned */
#define LACX_PCM_INTERLEAVED_I24 2u /* data0 = interleaved packed 3-byte little-endian samples */
typedef struct lacx_pcm {
    const void* data0;
    const void* data1;
    uint32_t layout;
    uint32_t channels; /* 1 or 2 */
} lacx_pcm;

/* Shard encode of device-resident PCM in any layout, device-side emit, zero-copy result (see
 * lacx_encode_shard_device_view).  The configured bit depth must match an interleaved layout. */
int lacx_encode_shard_pcm_device_view(lacx_encoder* enc, const lacx_pcm* d_pcm, uint64_t frames, void* stream,
                                      const uint8_t** payload, uint64_t* payload_size, const uint32_t** table,
                                      uint32_t* nblocks);

/* The same in two halves, for batch jobs (many files or shards through one process): _begin enqueues the whole
 * encode on the device and returns without waiting; _end waits for it and hands the result over.  One encode can be
 * in flight per encoder; with two encoders used alternately (begin A, end B, begin B, end A, ...) the device analyses
 * the next input while the previous one's last emit kernels are still pushing their payload over PCIe.  The result
 * views stay valid until the same encoder's next _begin. */
int lacx_encode_shard_pcm_device_begin(lacx_encoder* enc, const lacx_pcm* d_pcm, uint64_t frames, void* stream);
int lacx_encode_shard_end(lacx_encoder* enc, const uint8_t** payload, uint64_t* payload_size, const uint32_t** table,
                          uint32_t* nblocks);

/* Many streams as ONE job (BASELINE configs[4]: a mixed corpus; the reference keeps one pool over all blocks of a stream,
 * ref src/codec/lac/encoder.cpp:404-435 -- here the pool spans the blocks of all streams of the batch): one launch set
 * over every block of every stream instead of one per stream, so that short streams do not each pay the chain
 * ingest -> Levinson -> probes -> decision -> analysis by themselves.  Every stream keeps its own sample rate, bit depth,
 * channel count, stereo mode and layout; zero-run / partitioning switches come from the encoder's config.  Device-
 * resident PCM, device-side emit; out[i] views the stream's payload and block table inside the encoder's pinned result
 * buffer (valid until the next call on the encoder); lacx_assemble turns (payload, table) into the stream's .lac, the
 * bytes LAC::Encoder::encode gives for that stream alone.  Errors name the stream ("stream 3: left sample at index
 * ... is outside ..."). */
typedef struct lacx_batch_item {
    lacx_pcm pcm;         /* device-resident PCM of the stream */
    uint64_t frames;
    uint32_t sample_rate; /* 44100 / 48000 / 96000 / 192000 */
    uint8_t bit_depth;    /* 16 / 24 (an interleaved layout must match it) */
    uint8_t stereo_mode;  /* 0 LR, 1 MS, 2 per-block auto (ignored for mono) */
    uint8_t reserved[2];
} lacx_batch_item;
typedef struct lacx_batch_out {
    const uint8_t* payload;
    uint64_t payload_size;
    const uint32_t* table; /* (frames, bytes) per block */
    uint32_t nblocks;
    uint32_t reserved;
} lacx_batch_out;
int lacx_encode_batch_device(lacx_encoder* enc, const lacx_batch_item* items, uint32_t nstreams, void* stream,
                             lacx_batch_out* out);

/* WAV ingest (replaces read_wav + LAC::Encoder::encode of the CLI's encode command,
 * ref src/io/wav_io.cpp:167-277, src/main.cpp:640-675).  lacx_wav_parse walks the RIFF container in memory and
 * accepts / rejects exactly the files read_wav does (LACX_OK / LACX_E_INVALID, no device needed); lacx_encode_wav
 * copies the raw data chunk to the device as it is (2 or 3 bytes per sample), the kernels de-interleave and
 * sign-extend on load, and the complete .lac comes back -- the bytes the reference produces from the same file.
 * The encoder's sample_rate and bit_depth must match the file's. */
typedef struct lacx_wav_info {
    uint16_t channels;
    uint16_t bit_depth;
    uint32_t sample_rate;
    uint64_t frames;
    uint64_t data_offset; /* byte offset of the first sample in the file */
    uint64_t data_bytes;
} lacx_wav_info;
int lacx_wav_parse(const uint8_t* wav, uint64_t size, lacx_wav_info* out);
int lacx_encode_wav(lacx_encoder* enc, const uint8_t* wav, uint64_t size, uint8_t** out, uint64_t* out_size);
/* Zero-copy variant: *out points at the complete .lac inside the encoder's pinned result buffer (the device wrote the
 * payload there, header and block table are filled in in front of it); valid until the next call on the same encoder.
 * The upload is pipelined: the data chunk goes to the device in three pieces (1 : 3 : 4), each in front of its kernels. */
int lacx_encode_wav_view(lacx_encoder* enc, const uint8_t* wav, uint64_t size, const uint8_t** out, uint64_t* out_size);

/* Host-only: header + block table + payload concat of shards given in stream order. */
int lacx_assemble(const lacx_config* cfg, int channels, uint32_t nshards, const uint8_t* const* payloads,
                  const uint64_t* payload_sizes, const uint32_t* const* tables, const uint32_t* nblocks,
                  uint8_t** out, uint64_t* out_size);

/* ---- one stream over several devices (SURVEY 8(b) "multi-GPU fan-out lives entirely behind this shim", 8(e)) ----------
 * The reference's LAC::Encoder::encode spreads the blocks of a stream over its worker threads and concatenates their
 * payloads in block order (ref src/codec/lac/encoder.cpp:385-443 worker pool, :445-465 container).  An encoder created over
 * a device list does the same with devices as the workers: lacx_encode, lacx_encode_wav and lacx_encode_wav_view cut the
 * stream into contiguous block ranges [g*B/G, (g+1)*B/G) (lacx_fanout_range), one per lane; every lane has its own host
 * thread, streams, workspace and pinned result region on its device, uploads its range straight from the caller's buffer
 * and runs the single-device pipeline; the lanes exchange (payload bytes, block count) -- an RCCL all-gather of two u64
 * per lane over xGMI when the devices are distinct, a host-side sum otherwise (RCCL refuses two ranks on one device) --
 * and each lane copies its payload and its slice of the block table to its place in the final .lac.  The bytes do not
 * depend on the device list (blocks are independent).  Every other entry point of such an encoder runs on its first device.
 * devices: HIP ordinals, at most LACX_MAX_FANOUT; a device may appear more than once (a rehearsal of the fan-out on fewer
 * GPUs than lanes).  min_blocks_per_device: a stream is spread over fewer lanes when a lane would get fewer blocks than
 * this (0 = default 64; the reference uses min(threads, blocks) workers, encoder.cpp:385-390).
 * lacx_config.device = LACX_DEVICE_ALL in lacx_encoder_create is the list of every visible device.  LACX_FANOUT_EXCHANGE =
 * host | rccl (read at creation) forces the exchange. */
#define LACX_MAX_FANOUT 16u
int lacx_encoder_create_multi(const lacx_config* cfg, const int32_t* devices, uint32_t ndevices,
                              uint32_t min_blocks_per_device, lacx_encoder** out);
uint32_t lacx_encoder_lanes(const lacx_encoder* enc); /* 1 for a plain encoder */
/* The block range of lane `lane` of `nlanes` over a stream of `nblocks` blocks. */
void lacx_fanout_range(uint32_t nblocks, uint32_t nlanes, uint32_t lane, uint32_t* first, uint32_t* count);

/* Shards already resident in device memory, shards[g] on the device of lane g (every shard but the last a whole number of
 * blocks): every lane encodes its shard, the sizes are exchanged, out[g] views the lane's payload and block table in its
 * pinned result region (valid until the encoder's next call) with its byte offset in the stream's payload.  No
 * concatenation: lacx_assemble builds the .lac from the views where one contiguous buffer is wanted. */
typedef struct lacx_fanout_shard {
    lacx_pcm pcm; /* on the lane's device */
    uint64_t frames;
} lacx_fanout_shard;
typedef struct lacx_fanout_out {
    const uint8_t* payload;
    uint64_t payload_size;
    const uint32_t* table; /* (frames, bytes) per block */
    uint32_t nblocks;
    int32_t device;
    uint64_t byte_offset; /* of this payload inside the concatenated payload of the stream */
} lacx_fanout_out;
int lacx_encode_fanout_resident(lacx_encoder* enc, const lacx_fanout_shard* shards, uint32_t nshards, lacx_fanout_out* out);

#define LACX_EXCHANGE_HOST 1u
#define LACX_EXCHANGE_RCCL 2u
typedef struct lacx_fanout_stats {  /* of the encoder's last fanned-out call */
    uint32_t lanes_used;
    uint32_t exchange;               /* LACX_EXCHANGE_* */
    double exchange_ms;              /* longest lane: its shard finished -> every lane's sizes known (includes waiting for the slowest lane) */
    double concat_ms;                /* longest lane: table slice + payload copy into the final buffer */
    int32_t device[LACX_MAX_FANOUT];
    uint32_t blocks[LACX_MAX_FANOUT];
    uint64_t lane_frames[LACX_MAX_FANOUT];
    uint64_t payload_bytes[LACX_MAX_FANOUT];
    double encode_ms[LACX_MAX_FANOUT]; /* the lane's shard: upload, kernels, payload in its pinned region */
} lacx_fanout_stats;
int lacx_get_fanout_stats(const lacx_encoder* enc, lacx_fanout_stats* out);
int lacx_get_lane_timing(const lacx_encoder* enc, uint32_t lane, lacx_timing* out);
const char* lacx_fanout_exchange_note(const lacx_encoder* enc); /* which exchange the encoder uses, and why */

/* ---- decode: LAC::Decoder::decode, ref src/codec/lac/decoder.hpp:10-24, decoder.cpp:76-303,
 * src/codec/block/decoder.cpp:64-520.  The product's own check that a .lac gives back the PCM, on the device: one lane
 * per block (the format serialises everything inside a block), all blocks of the stream at once.
 * lacx_stream_parse: host only; the reference reader's structural rules for the header and the block table (versions 3
 * and 2), LACX_E_INVALID otherwise.  Two documented deviations from the reference reader: its 1 GiB cap on the decoded PCM
 * is not taken over (it would refuse the 2 h stream), and a compressed block must stay below 2^29 bytes (the device
 * reader's bit positions are 32-bit and relative to the block; the reference accepts any non-zero size that fits the
 * file -- no encoder produces such a block: 16384 frames x 2 channels cost at most a few hundred KiB).
 * lacx_decode: left / right (right may be null for mono) are caller-owned arrays of `frames` int32 each; a malformed
 * block, a sample outside the bit depth or a residual magnitude the encoder's domain cannot produce (>= 2^30) gives
 * LACX_E_RUNTIME ("[decode-error] block=N ..." in lacx_decode_last_error, the reference throws std::runtime_error with
 * that prefix, decoder.cpp:24-32).  device = -1: the current device.  device_ms (nullable): kernel time. */
typedef struct lacx_stream_info {
    uint32_t sample_rate;
    uint32_t blocks;
    uint64_t frames;
    uint8_t channels;
    uint8_t bit_depth;
    uint8_t stereo_mode;
    uint8_t version; /* 3, or the legacy 2 (no compressed block sizes: decoded by one lane) */
} lacx_stream_info;
int lacx_stream_parse(const uint8_t* lac, uint64_t size, lacx_stream_info* out);
int lacx_decode(int device, const uint8_t* lac, uint64_t size, int32_t* left, int32_t* right, uint64_t frames,
                float* device_ms);
const char* lacx_decode_last_error(void); /* of the calling thread */
/* The same through a decoder object (ref LAC::Decoder, src/codec/lac/decoder.hpp:10-24) whose device buffers, stream and
 * events live from call to call; lacx_decode keeps one such object per calling thread and device.  device = -1: the device
 * that is current at the first call. */
typedef struct lacx_decoder lacx_decoder;
int lacx_decoder_create(int device, lacx_decoder** out);
void lacx_decoder_destroy(lacx_decoder* dec);
int lacx_decoder_decode(lacx_decoder* dec, const uint8_t* lac, uint64_t size, int32_t* left, int32_t* right, uint64_t frames,
                        float* device_ms);

/* Block::Encoder::encode drop-in for one channel block of n <= 16384 samples of ANY int32 value: blocks inside the 25-bit
 * mid/side domain of validated 16 / 24-bit input run on the streaming kernels, wider ones on a kernel of their own that
 * follows the reference through its int32-overflow order fallback (ref lpc.cpp:24-36, 188-229) and up to k = 31.
 * More than 16384 samples: LACX_E_INVALID (the LAC container cannot carry such a block). */
int lacx_block_encode(lacx_encoder* enc, const int32_t* pcm, uint32_t n, uint8_t** out, uint64_t* out_size);
int lacx_block_plan_only(lacx_encoder* enc, const int32_t* pcm, uint32_t n, lacx_channel_plan* plan);

/* Kernel-level probes for parity tests: exact autocorrelation + Q15 candidate sets of one channel
 * block as the device computes them. acorr[13]; coef[5*13]; used[5]. */
int lacx_debug_lpc(lacx_encoder* enc, const int32_t* pcm, uint32_t n, int64_t* acorr, int16_t* coef,
                   uint8_t* used);

/* Diagnostic builds (-DLACX_STAMPS) only: per-phase shader-cycle sums of k_analyze<16,1024>; returns 0 in
 * production builds. out[40]; out[32] = number of waves accumulated. */
int lacx_debug_stamps(unsigned long long* out32);

/* Host-side worker threads the encoder's emit pool runs besides the calling thread (creates the pool; no device
 * needed).  emit_threads = 1 must give 0: the reference's set_thread_count(1) means one thread in total
 * (ref src/codec/lac/encoder.cpp:385-390). */
int lacx_debug_emit_workers(lacx_encoder* enc);

/* Number of visible HIP devices (0 when the runtime or a GPU is missing); does not initialise one. */
int lacx_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
