/*
 * atsc_hip.h -- C-ABI of the MI355X-native ATSC 8VSB modulator hot path.
 *
 * This is the drop-in boundary for the encoder library of philburr/atsc
 * (reference: src/atsc/atsc.h:7-16 -- class atsc::atsc_encoder { create(); process(pkt,count,fn); }).
 * Everything behind these entry points runs as hand-written HIP kernels for gfx950; there is no
 * CPU fallback: every call fails with ATSC_HIP_ERR_NO_DEVICE / a HIP error when no GPU is present.
 *
 * Plain C types only: pointers, sizes, ints.  `stream` arguments are hipStream_t passed as void*
 * (NULL = the default stream).  Device pointers are marked d_*, host pointers h_*.
 *
 * A handle ("encoder") owns `channels` independent TS streams that advance in lockstep, one field
 * index for all of them (they share one oscillator table, reference: src/atsc/offset.h:9,81).
 * Like the reference a handle is NOT thread-safe; handles are independent.
 */
#ifndef ATSC_HIP_H
#define ATSC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Sizes of the path (reference: src/common/atsc_parameters.h:5-25). */
#define ATSC_HIP_TS_PACKET_BYTES 188
#define ATSC_HIP_PACKETS_PER_FIELD 312
#define ATSC_HIP_FIELD_TS_BYTES (312 * 188)      /* 58656   */
#define ATSC_HIP_FIELD_CODED_BYTES (312 * 207)   /* 64584   */
#define ATSC_HIP_FIELD_SYMBOLS (313 * 832)       /* 260416  */
#define ATSC_HIP_FIELD_IQ_BYTES (313 * 832 * 8)  /* 2083328 */
#define ATSC_HIP_PADDED_SYMBOLS (314 * 832)      /* 261248: oscillator table length, offset.h:9 */
#define ATSC_HIP_INTERLEAVER_TAIL 10608          /* 208*51 coded bytes carried across a field boundary */

/* Return codes: 0 ok; >0 a hipError_t value; <0 the codes below. */
#define ATSC_HIP_OK 0
#define ATSC_HIP_ERR_ARG (-1)
#define ATSC_HIP_ERR_NO_DEVICE (-2)
#define ATSC_HIP_ERR_TOO_LARGE (-3)
#define ATSC_HIP_ERR_STATE (-4)

typedef struct atsc_hip_encoder atsc_hip_encoder;

typedef struct atsc_hip_config {
    int device;      /* HIP device ordinal */
    int channels;    /* independent TS streams encoded in lockstep (>=1) */
    int max_fields;  /* largest number of fields per channel in one batch (>=1) */
    int flags;       /* ATSC_HIP_FLAG_* */
} atsc_hip_config;

#define ATSC_HIP_FLAG_TIMING 1 /* record HIP events around every kernel of a batch (atsc_hip_last_timings) */
#define ATSC_HIP_FLAG_TAPS 2   /* also materialise the RS-coded byte stream of every batch (atsc_hip_stage_buffer
                                  ATSC_HIP_STAGE_RAND_RS); the path itself goes TS + parity -> interleaver directly */

#define ATSC_HIP_FLAG_STRICT_FP32 4 /* the reference's arithmetic only (src/atsc/filter.h:16-69 is fp32 FMA): AUTO never picks a matrix-core
                                  form, atsc_hip_set_fir_form(MFMA / FUSED) and atsc_hip_set_upsample(2) return ATSC_HIP_ERR_ARG.  The
                                  IQ bits of such a handle do not depend on its channel count (see atsc_hip_get_fir_form) */

/* Stage ids for atsc_hip_stage_buffer / atsc_hip_last_timings. */
enum {
    ATSC_HIP_STAGE_RAND_RS = 0,   /* randomizer + RS(207,187)            -> coded bytes   */
    ATSC_HIP_STAGE_INTERLEAVE,    /* convolutional byte interleaver      -> interleaved   */
    ATSC_HIP_STAGE_TRELLIS_PAR,   /* per-part precoder/trellis parities                   */
    ATSC_HIP_STAGE_TRELLIS_SCAN,  /* exclusive XOR scan of the parities  -> carries       */
    ATSC_HIP_STAGE_TRELLIS_MAP,   /* trellis + 8-level map + sync insert -> u8 symbols    */
    ATSC_HIP_STAGE_OSC,           /* oscillator table of the batch (or the bare state advance)   */
    ATSC_HIP_STAGE_MIX_FIR,       /* pilot/offset mixer + RRC FIR        -> complex64 IQ  */
    ATSC_HIP_STAGE_STATE,         /* end-of-batch state save                               */
    ATSC_HIP_NSTAGES
};

/* Per-field completion callback of the streaming path.  Replaces the reference's
 * std::function<void(void*,unsigned)> (src/atsc/atsc.h:11): same arguments plus a user pointer.
 * `iq` is library-owned pinned host memory, valid only during the callback (atsc.cpp:89); `bytes` follows the
 * handle's sample format (atsc_hip_set_iq_format). */
typedef void (*atsc_hip_field_cb)(void* user, void* iq, unsigned bytes);

/* Replaces atsc::atsc_encoder::create() (src/atsc/atsc.cpp:93-95).  Allocates all device buffers
 * for channels x max_fields, builds the constant tables, zero history (atsc.cpp:17-35). */
int atsc_hip_create(const atsc_hip_config* cfg, atsc_hip_encoder** out);

/* Replaces ~atsc_encoder (src/atsc/atsc.h:9). */
void atsc_hip_destroy(atsc_hip_encoder* enc);

/* Replaces atsc::atsc_encoder::process (src/atsc/atsc.cpp:59-77) for a 1-channel handle:
 * `count` contiguous 188-byte TS packets (host memory, copied before return); every completed
 * 312-packet field fires cb(user, iq, bytes) synchronously, in order, before the call returns; bytes = 260416 x the
 * handle's sample size: 2083328 for CF32 (the reference's payload), 1041664 for CS16, 520832 for CS8; twice that on a handle with atsc_hip_set_upsample(2).
 * All complete fields available are encoded in GPU batches of up to max_fields; when a call completes several
 * batches, the upload and kernels of batch k+1 overlap the download and callbacks of batch k (two internal slots
 * and streams) -- nothing stays in flight when the call returns.  A trailing partial field stays buffered (no flush
 * API, like the reference).
 * Stream order: the call runs on streams of its own; it first waits (on the device) for whatever the last batch call or
 * atsc_hip_seek on this handle queued on the caller's stream, and returns with its own streams drained, so batch and
 * streaming calls on one handle may be mixed from one host thread without a synchronisation in between.
 * Errors: packets that were handed to the GPU before a failure are dropped from the buffer (they have advanced the
 * stream state and are not encoded twice); after any error return from a process / encode call the handle's stream
 * state (interleaver history, trellis state, oscillator) is undefined until atsc_hip_import_state re-seats it.  A
 * failed atsc_hip_encode_finish abandons the pending batch; the handle accepts new batches afterwards. */
int atsc_hip_process(atsc_hip_encoder* enc, const uint8_t* h_pkts, unsigned count, atsc_hip_field_cb cb, void* user);

/* Batched multi-channel path behind process_field (src/atsc/atsc.cpp:79-90), device-resident.
 *   d_ts  : [channels][fields][312][188] u8 TS packets
 *   d_sym : [channels][fields][260416]  u8 level indices 0..7 (symbol tap after field_sync,
 *           atsc.cpp:84), or NULL to use the library's internal symbol buffer
 *   d_iq  : [channels][fields][260416]  complex64 (re,im fp32) baseband (atsc.cpp:86,89) -- or the handle's fixed-point
 *           sample format, atsc_hip_set_iq_format -- or NULL to stop after the symbols (the oscillator still
 *           advances `fields` fields)
 * Asynchronous on `stream`; 1 <= fields <= max_fields. */
int atsc_hip_encode_batch(atsc_hip_encoder* enc, const uint8_t* d_ts, int fields, uint8_t* d_sym, void* d_iq,
                          void* stream);

/* The same batch in two halves, for the single-stream time-split case (BASELINE config 5).
 * No reference counterpart (the reference is strictly field-serial); callers: multi-GPU hosts that split
 * ONE stream at field boundaries.
 *   begin : stages that do not depend on the trellis state entering the batch (randomizer, RS, interleaver,
 *           per-part parities); writes the span parity per channel to d_span_parity (16 B per channel: three
 *           little-endian words for encoders 0-3 / 4-7 / 8-11, one byte per encoder with bit0 = precoder
 *           delay, bit1 = D1, bit2 = D2 contribution; fourth word 0), or skips it when NULL.
 *   finish: d_carry_xor (same 16 B format per channel, device memory, or NULL) is XORed into the handle's own
 *           trellis state -- for rank r of a time-split: the XOR of the span parities of all earlier ranks --
 *           then scan, trellis/map/sync, mixer+FIR, state save.
 * encode_batch == begin(NULL) + finish(NULL).  Everything stays asynchronous on `stream`. */
int atsc_hip_encode_begin(atsc_hip_encoder* enc, const uint8_t* d_ts, int fields, void* d_span_parity, void* stream);
int atsc_hip_encode_finish(atsc_hip_encoder* enc, const void* d_carry_xor, uint8_t* d_sym, void* d_iq, void* stream);

/* SDR sink adapter.  The reference hands the radio CF32 only (util/atsc_encoder.cpp:75,
 * SOAPY_SDR_CF32); most SDR wire formats are fixed point, and 8 B/sample is what bounds the host link.  The sample
 * format of every d_iq / h_iq / callback buffer of a handle can therefore be switched; the conversion is fused into
 * the FIR kernel's store (no extra pass over the IQ):
 *   CF32: interleaved re,im fp32 (8 B/sample) -- the reference's payload, default
 *   CS16: interleaved re,im int16 (4 B/sample), q = rint(clamp(y,-1,1) * 32767), ties to even (SOAPY_SDR_CS16 scaling)
 *   CS8 : interleaved re,im int8  (2 B/sample), q = rint(clamp(y,-1,1) * 127),   ties to even (SOAPY_SDR_CS8)
 * y is the CF32 value of the same encode (full scale 1.0, observed peak 0.77).  No reference counterpart for the fixed-
 * point formats: parity is pinned THROUGH the CF32 output (q must equal the formula applied to it). */
enum { ATSC_HIP_IQ_CF32 = 0, ATSC_HIP_IQ_CS16 = 1, ATSC_HIP_IQ_CS8 = 2 };
int atsc_hip_set_iq_format(atsc_hip_encoder* enc, int format);
size_t atsc_hip_iq_sample_bytes(int format); /* 8 / 4 / 2; 0 for an unknown format */

/* Which kernel computes offset.process_field + filter.process_field (src/atsc/atsc.cpp:85-86).  Both are parity-checked
 * against the same oracle to the same tolerance (1e-5 of full scale); their outputs agree to ~2e-6 of full scale but are not
 * bit-identical.  FFT is fp32 throughout (2.4e-7 against the oracle); the matrix-core forms multiply fp16 pieces of both
 * operands with fp32 accumulation: 1.7e-6 measured, 7.2e-6 worst case over every input with every error term at its
 * bound (fir_fused.hip, scripts/fir_error_budget.py).
 *   FFT : overlap-save fast convolution on the vector units (any channel count)
 *   MFMA: banded GEMM on the matrix cores; needs the oscillator shared by many channels to pay off
 *   FUSED: the MFMA form with the trellis / map / sync stage (src/atsc/atsc.cpp:83-84) fused in: the symbols are
 *         regenerated from the interleaved bytes inside the kernel and never pass through HBM.  Only in
 *         atsc_hip_encode_batch / _finish with d_iq != NULL; d_sym is then an optional tap (NULL: no symbols are
 *         written anywhere and atsc_hip_stage_buffer(TRELLIS_MAP) returns ATSC_HIP_ERR_STATE)
 *   WIDE: FUSED with the oscillator folded into the TAP operand instead of the sample operand (fir_w.hip): the levels are
 *         exact in one fp16 piece, so a third fewer matrix products, shared by 64 channels per wave; complex64 at one sample
 *         per symbol and a channel count that is a multiple of 64 only (ATSC_HIP_ERR_ARG otherwise); 1.2e-6 measured (the
 *         lowest of the matrix-core forms), 8.4e-6 worst case with every term at its bound, 7 % more time than FUSED on
 *         64 channels x 64 fields and 6 % less board power.  Never AUTO's choice; kept as the measured alternative.
 *   AUTO: FUSED when the handle's
 *         channels fill at least 80 % of whole groups of 16 (13..16, 26..32, 39..48, 52..),
 *         FFT otherwise (default)
 * MFMA and FUSED address 16 channel rows of a batch's IQ through one 32-bit-offset buffer descriptor: they need
 * 16 * max_fields * 260416 * atsc_hip_iq_sample_bytes(format) < 4 GiB (max_fields <= 128 for CF32, 257 for CS16, 515
 * for CS8).  On a larger handle AUTO resolves to FFT, and asking for MFMA / FUSED (or switching such a handle to a wider
 * sample format) returns ATSC_HIP_ERR_TOO_LARGE.
 * No reference counterpart (the reference has one scalar/AVX form, filter.h:16-69). */
enum { ATSC_HIP_FIR_AUTO = 0, ATSC_HIP_FIR_FFT = 1, ATSC_HIP_FIR_MFMA = 2, ATSC_HIP_FIR_FUSED = 3, ATSC_HIP_FIR_WIDE = 4 };
int atsc_hip_set_fir_form(atsc_hip_encoder* enc, int form);
/* The form the handle's next batch runs (AUTO resolved: a function of the channel count and the setting only), or a
 * negative error code.  The forms differ in the last bits of the IQ, so the ranks of a time-split must
 * all run the same one for the spans to join bit-identically: atsc_amd/timesplit.py compares this value across ranks.
 * REPRODUCIBILITY: under AUTO the IQ bits of a stream therefore depend on how many channels share its handle (a 12-channel
 * handle runs FFT, a 13-channel one FUSED); every form is within the 1e-5 tolerance of the reference, none is bit-identical
 * to another.  Callers that need the same bits whatever the grouping pin the form (atsc_hip_set_fir_form) or create the
 * handle with ATSC_HIP_FLAG_STRICT_FP32. */
int atsc_hip_get_fir_form(atsc_hip_encoder* enc);

/* Interpolating form of the filter (the "upsampling FIR to the SDR rate"): factor 2 makes every d_iq / h_iq /
 * callback buffer hold 2 x 260416 samples per field,
 *     y2[2 i]     = sum_k taps[k]      * z[i + k]      the reference's sample (src/atsc/filter.h:16-69), bit for bit the 1x output
 *     y2[2 i + 1] = sum_k taps_half[k] * z[i + k]      the same root-raised-cosine pulse, same scale, half a symbol later
 * (taps_half: 112 taps symmetric about k = 55.5, atsc_hip_get_table), computed in the SAME pass over the symbols by the
 * matrix-core kernel: a polyphase interpolator by 2 whose phase 0 is the reference filter -- no 1x IQ in HBM, no second
 * pass.  No reference counterpart (util/atsc_encoder.cpp:70 runs the radio at the symbol rate): the odd samples are
 * PARITY UNPINNED, checked against a float64 model of the formula and by the spectrum (tests/test_gpu_upsample.py); the
 * even samples are the reference's.  A 2x handle runs MFMA / FUSED only (AUTO -> FUSED whatever the channel count;
 * atsc_hip_set_fir_form(FFT) returns ATSC_HIP_ERR_ARG) and needs 16 * max_fields * 260416 * 2 * sample bytes < 4 GiB
 * (max_fields <= 64 for CF32; ATSC_HIP_ERR_TOO_LARGE otherwise).  Other ratios: atsc_hip_resample.h (a second pass). */
int atsc_hip_set_upsample(atsc_hip_encoder* enc, int factor); /* 1 (default) or 2 */
int atsc_hip_get_upsample(atsc_hip_encoder* enc);

/* K streams through the reference's process() contract on ONE K-channel handle.  Replaces K instances of
 * atsc::atsc_encoder (src/atsc/atsc.h:7-16, one per TS as in util/atsc_encoder.cpp:83-100), each with its own
 * process(pkt, count, fn) (src/atsc/atsc.cpp:59-77): `channel` says which stream the packets belong to.  The packets are
 * copied before the call returns and (cb, user) is remembered as THAT stream's callback.  Whenever every stream of the
 * handle has at least one whole field buffered, the call that completes the set runs the fields all streams have
 * (batches of up to max_fields, two slots: upload + kernels of batch k+1 overlap the download and callbacks of batch k)
 * through atsc_hip_encode_batch -- the many-channel kernels, the fused matrix-core form when the channel count fills its
 * groups (atsc_hip_get_fir_form) -- and fires every stream's callback for its fields: in field order per stream,
 * field-major across streams, synchronously on the thread of THIS call, before it returns.  So a stream's callback may
 * run inside another stream's call; it must stay callable until the handle is destroyed or the stream's next call
 * replaces it.  Nothing stays in flight between calls; a trailing partial field stays buffered.
 * The streams advance in lockstep (one field index for the handle): a stream that ends starves the others -- keep feeding
 * it null packets, as a multiplexer does.  A stream may be ahead of the slowest one by at most
 * ATSC_HIP_MULTI_MAX_AHEAD x max_fields fields (ATSC_HIP_ERR_TOO_LARGE beyond; nothing is copied then).
 * Not thread-safe, like the reference (one thread at a time per handle).  Cannot be mixed with atsc_hip_process on the
 * same handle while either has bytes buffered (ATSC_HIP_ERR_STATE).  Errors as for atsc_hip_process.
 * Host memory: 2 x channels x max_fields x (58656 + 260416 x sample bytes x upsample) pinned bytes, allocated on first use. */
#define ATSC_HIP_MULTI_MAX_AHEAD 8
int atsc_hip_process_multi(atsc_hip_encoder* enc, int channel, const uint8_t* h_pkts, unsigned count, atsc_hip_field_cb cb,
                           void* user);
/* TS packets of `channel` buffered by atsc_hip_process_multi and not yet encoded.  No reference counterpart (the
 * reference's field buffer is private, src/atsc/atsc.cpp:61-70); for callers that pace the streams. */
int atsc_hip_multi_pending(atsc_hip_encoder* enc, int channel, size_t* packets);

/* atsc_hip_process with the radio-side chunking of the reference's transmitter loop (util/atsc_encoder.cpp:84-100:
 * `transfer = min(remaining, MTU)` per writeStream call, chunks never span a field): cb(user, samples, n, format)
 * for every piece of at most mtu_samples samples of every completed field, synchronously and in order. */
typedef void (*atsc_hip_chunk_cb)(void* user, const void* samples, size_t nsamples, int format);
int atsc_hip_process_chunked(atsc_hip_encoder* enc, const uint8_t* h_pkts, unsigned count, size_t mtu_samples,
                             atsc_hip_chunk_cb cb, void* user);

/* Same, from/to host memory (pageable or pinned); synchronous.  h_sym / h_iq may be NULL. */
int atsc_hip_encode_batch_host(atsc_hip_encoder* enc, const uint8_t* h_ts, int fields, uint8_t* h_sym, void* h_iq);

/* Mixer + FIR alone on caller-supplied symbols (reference: offset.process_field + filter.process_field,
 * src/atsc/atsc.cpp:85-86).  Advances the oscillator by `fields`. */
int atsc_hip_mix_fir(atsc_hip_encoder* enc, const uint8_t* d_sym, int fields, void* d_iq, void* stream);

/* The same mixer + FIR with NO handle and NO allocation.  No counterpart in the reference; it is for callers that cannot
 * allocate: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Everything the kernels
 * need beyond d_sym and d_iq (the oscillator table T_0, the oscillator state and the per-field table of the batch) lives
 * in one fixed device array of the library (14.6 MB per GPU, held until the process ends), so the batch is limited to
 * ATSC_HIP_STATIC_MAX_FIELDS fields per channel (ATSC_HIP_ERR_TOO_LARGE beyond; any number of channels).
 *   d_sym : [channels][fields][260416] u8 level indices;  d_iq : [channels][fields][260416] samples in `iq_format`
 *   field_index : absolute index of the batch's first field (T_f = T_0 (x) scale^f is re-derived from T_0 on every call,
 *                 by the reference's fp32 recursion, offset.h:35-41); at most 2^24
 * The first call on a device sets that device up (host tables, two copies to the device): make it outside any graph
 * capture, e.g. with fields = 0, which does the set-up and nothing else.  After that a call allocates nothing on the
 * device or the host and is asynchronous on `stream`.  All calls on one GPU share the one array: calls that overlap on a
 * GPU (different host threads, or streams with no order between them) race; results equal atsc_hip_mix_fir's (FFT form)
 * bit for bit. */
#define ATSC_HIP_STATIC_MAX_FIELDS 4
int atsc_hip_mix_fir_static(int device, const uint8_t* d_sym, int channels, int fields, uint64_t field_index, void* d_iq,
                            int iq_format, void* stream);

/* Stream state at a field boundary: everything field f+1 needs from the past.
 * Reference counterparts: interleaver next_field contents (atsc.cpp:53-54,82,88), trellis carries
 * (trellis.h:500-502), saved_symbols (atsc.cpp:37, field_sync.h:17,23), field parity
 * (field_sync.h:10,25), oscillator table phase (offset.h:9,40-41). */
typedef struct atsc_hip_stream_state {
    uint64_t field_index;                               /* fields encoded so far (shared by all channels) */
    uint8_t coded_tail[ATSC_HIP_INTERLEAVER_TAIL];      /* last 10608 RS-coded bytes of the previous field */
    uint8_t last_group[12];                             /* last 12 interleaved bytes of the previous field  */
    uint8_t z2[12];                                     /* precoder delay per trellis encoder               */
    uint8_t z0_next[12];                                /* D1: Z0 of the next symbol                        */
    uint8_t z0_prev[12];                                /* D2: Z0 of the last symbol                        */
} atsc_hip_stream_state;

int atsc_hip_export_state(atsc_hip_encoder* enc, int channel, atsc_hip_stream_state* out);
/* Import sets the channel history; field_index moves the handle's field counter and re-derives the
 * oscillator table T_f = T_0 (x) scale^f by the reference's fp32 recursion when it changes. */
int atsc_hip_import_state(atsc_hip_encoder* enc, int channel, const atsc_hip_stream_state* in);

/* Re-seat EVERY channel of the handle at absolute field `field_index` with a zero trellis / precoder state, on the device
 * and in stream order: no host round trip, no synchronisation.  The interleaver history the handle holds (coded tail, last
 * interleaved group: what the previous batch left) stays; the field parity follows the index and the oscillator state
 * becomes T_field_index by the reference's own fp32 recursion (offset.h:35-41: advanced from the handle's current state
 * when field_index lies ahead of it, from T_0 otherwise -- the same sequence of roundings either way).
 * This is the time-split's "start a span here" (SURVEY.md 8e): encode the one-field halo, seek to the span's first field,
 * then atsc_hip_encode_begin / _finish with the carry of the earlier spans.  It does for all channels at once what an
 * atsc_hip_export_state / atsc_hip_import_state pair per channel did through the host.  No reference counterpart: the
 * reference cannot start mid-stream (atsc.cpp:34-40 zero state only).
 * Returns ATSC_HIP_ERR_STATE between atsc_hip_encode_begin and _finish.
 * The oscillator walk is one dependent fp32 step per field: at most ATSC_HIP_MAX_SEEK_STEPS of them per call (counted from
 * the handle's current index when the target lies ahead, from 0 otherwise), ATSC_HIP_ERR_TOO_LARGE beyond; the same bound
 * applies to the field_index of atsc_hip_import_state.  The handle's host-side state moves only if every launch was accepted. */
#define ATSC_HIP_MAX_SEEK_STEPS ((uint64_t)1 << 24)
int atsc_hip_seek(atsc_hip_encoder* enc, uint64_t field_index, void* stream);

/* Trellis parity of a span of fields encoded from zero trellis state: out36[0..11]=z2, [12..23]=z0_next,
 * [24..35]=z0_prev contribution.  state_out = state_in XOR span_parity.  Valid after a batch. */
int atsc_hip_span_parity(atsc_hip_encoder* enc, int channel, uint8_t* out36);

/* Device pointer + byte size of the internal buffer written by a stage in the last batch (stage taps for
 * the parity tests: coded bytes, interleaved bytes, symbols).  Layouts are documented in DESIGN.md. */
int atsc_hip_stage_buffer(atsc_hip_encoder* enc, int stage, void** d_ptr, size_t* bytes, size_t* channel_stride);

/* Per-stage kernel milliseconds of the last batch (needs ATSC_HIP_FLAG_TIMING; synchronizes).
 * No reference counterpart (the reference has no tracing): measurement support. */
int atsc_hip_last_timings(atsc_hip_encoder* enc, float* ms /* [ATSC_HIP_NSTAGES] */);
/* Reserve a ring of `nslots` HIP event sets and restart batch numbering at 0, so that every batch of a
 * timed region keeps its own events (recorded on the stream the kernels are launched on) ... */
int atsc_hip_timing_reserve(atsc_hip_encoder* enc, int nslots);
/* ... and read batch `batch` (0-based since the last reserve; must still be in the ring) afterwards. */
int atsc_hip_batch_timings(atsc_hip_encoder* enc, int batch, float* ms /* [ATSC_HIP_NSTAGES] */);
/* Which stages the following batches time (bit i = ATSC_HIP_STAGE_i; ~0u = all, the default).  An event between two kernels
 * costs the stream about 5 us of idle time on this part, eight of them 1 % of a 64 x 64 step: a timed region that needs one
 * kernel's duration selects that stage alone (its mark and the one behind it are recorded), the other stages read 0.  The
 * fused path launches the carry scan, the oscillator table and the state save as ONE kernel, reported under
 * ATSC_HIP_STAGE_OSC.  No reference counterpart (measurement). */
int atsc_hip_timing_select(atsc_hip_encoder* enc, unsigned stage_mask);

/* Constant tables as the library built them (host copies), for table parity tests:
 * which: 0 = 112 FIR taps (float), 1 = oscillator T0 (261248 complex64), 2 = scale (2 floats),
 *        3 = PRBS bytes (58344), 4 = field sync even (820 B), 5 = field sync odd (820 B),
 *        6 = RS generator g[0..20], 7 = the 112 half-symbol taps of the 2x form (float),
 *        8 = the matrix-core FIR's tap operand, fp16 bit patterns [phase 2][hi, lo][K-step 9][lane 64][8] (36864 B),
 *        9 = the wide form's tap operand, fp16 bit patterns [hi, lo][K-step 8][lane 64][8] (16384 B).
 * enc may be NULL (tables do not depend on the device; no GPU needed); for 8 and 9 a handle reads the operand back from
 * its device, which is what the kernels multiply by (the error-budget tests use those bits). */
int atsc_hip_get_table(atsc_hip_encoder* enc, int which, void* h_out, size_t bytes);

const char* atsc_hip_error_string(int code);
const char* atsc_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ATSC_HIP_H */
