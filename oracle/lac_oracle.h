/* oracle/lac_oracle.h -- TEST INFRASTRUCTURE ONLY (parity checker + CPU baseline), never the product.
 *
 * Plain-C restatement of the reference's block-encode path (audexdev/Lossless-Audio-Codec, checked out
 * read-only at /root/reference).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library.  Parity status: PINNED -- byte-compared against the unmodified reference
 * compiled in place (oracle/_ref, see oracle/Makefile) by tests/test_oracle_vs_ref.py, and against the
 * golden fixtures under tests/golden/ minted from that reference build.
 */
#ifndef LAC_ORACLE_H
#define LAC_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LACO_MAX_BLOCK 16384u
#define LACO_MAX_PARTS 256u

/* Per-channel-block decisions, i.e. everything Block::Encoder::encode decides before emitting bits
 * (reference src/codec/block/encoder.cpp:313-552). */
typedef struct laco_plan {
    uint8_t predictor_type;           /* 0 fixed, 1 FIR, 2 LPC            encoder.cpp:54-56 */
    uint8_t order;                    /* chosen_order                      encoder.cpp:421-423 */
    uint8_t partition_order;          /* best_partition_order              encoder.cpp:481-552 */
    uint8_t reserved;
    int16_t coeffs_q15[13];           /* [1..order] valid for LPC          lpc.cpp:176-183 */
    uint16_t reserved2;
    uint32_t part_count;
    uint64_t total_bits;              /* best_total_bits (metadata + residual bits, byte padded) */
    uint64_t best_bits;               /* best.best_bits of the winning predictor */
    uint8_t part_mode[LACO_MAX_PARTS]; /* 0 rice 1 zero-run 2 bin 3 static */
    uint8_t part_k[LACO_MAX_PARTS];
} laco_plan;

typedef struct laco_stereo {
    int choose_ms;
    int uncertain;
    uint64_t sums[12]; /* l_raw,r_raw,m_raw,s_raw, l,r,m,s (diff), l,r,m,s (anti)  lac/encoder.cpp:130-141 */
} laco_stereo;

void laco_free(void* p);

/* LAC::Encoder::encode (src/codec/lac/encoder.cpp:215-466). Returns 0 ok, 1 invalid argument,
 * 2 runtime error. `right` NULL => mono. */
int laco_encode(const int32_t* left, const int32_t* right, uint64_t frames, uint32_t sample_rate,
                int bit_depth, int stereo_mode, int zero_run, int partitioning, int threads,
                uint8_t** out, uint64_t* out_size);

/* Block::Encoder::encode (src/codec/block/encoder.cpp:313-838). */
int laco_block_encode(const int32_t* pcm, uint32_t n, int zero_run, int partitioning, uint8_t** out,
                      uint64_t* out_size);

/* Same analysis, returning the decisions instead of the bytes. */
int laco_block_plan(const int32_t* pcm, uint32_t n, int zero_run, int partitioning, laco_plan* plan);

/* LPC::autocorrelation (src/codec/lpc/lpc.cpp:80-96), exact int64, lags 0..order. */
void laco_autocorr(const int32_t* pcm, uint32_t n, int order, int64_t* r);

/* LPC::analyze_block_q15 (src/codec/lpc/lpc.cpp:156-186). coeffs has order+1 entries. Returns used_order. */
int laco_lpc_analyze(const int32_t* pcm, uint32_t n, int order, int16_t* coeffs_q15);

/* Levinson on given autocorrelation (lags 0..order) -> Q15 coefficients; returns used_order. */
int laco_levinson_q15(const int64_t* r, int order, int16_t* coeffs_q15);

/* estimate_stereo_mode (src/codec/lac/encoder.cpp:126-197). */
void laco_stereo_estimate(const int32_t* left, const int32_t* right, uint32_t n, laco_stereo* out);

/* Rice::adapt_k sequence (src/codec/rice/rice.hpp:45-114): k_out[i] = k after value i. */
void laco_adapt_k_sequence(const uint32_t* u, uint32_t n, uint32_t* k_out);

/* LAC::Decoder::decode for v3 streams (src/codec/lac/decoder.cpp:76-303, block/decoder.cpp:64-520).
 * Returns 0 ok. left/right malloc'd (right NULL for mono). */
int laco_decode(const uint8_t* data, uint64_t size, int32_t** left, int32_t** right, uint64_t* frames,
                int* channels, uint32_t* sample_rate, int* bit_depth, int* stereo_mode);

#ifdef __cplusplus
}
#endif
#endif
