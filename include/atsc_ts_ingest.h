/*
 * atsc_ts_ingest.h -- TS ingest front-end of the MI355X 8VSB modulator: the step BEFORE the hot path.
 *
 * The reference has no such stage: util/atsc_encoder_null.cpp:38-43 and util/atsc_encoder.cpp:81-83 fread whole
 * 312-packet fields and trust that byte 0 of every 188 is the sync byte (randomize.h:48 drops it unchecked); a trailing
 * partial field is silently lost and a source slower than 19.39 Mbit/s starves the radio.  This front-end gives the
 * callers of atsc_hip_process / atsc_hip_encode_batch what they need to feed real streams:
 *   - 0x47 sync acquisition on an arbitrary byte stream (lock after `lock_packets` consecutive syncs 188 bytes apart),
 *     sync-loss detection (`loss_packets` consecutive bad syncs) and re-acquisition;
 *   - null-packet stuffing (PID 0x1FFF: 47 1F FF 10 + 184 x FF) to keep the fixed ATSC mux rate when the source
 *     underruns at a field deadline, and to complete the last field at end of stream.
 * Host-side byte parsing, plain C ABI, no GPU involved (it runs while the GPU encodes the previous batch).
 * Parity: no reference counterpart -> behaviour is specified here and tested against a Python model
 * (tests/test_ts_ingest.py); on a clean aligned stream the output is the input, byte for byte.
 */
#ifndef ATSC_TS_INGEST_H
#define ATSC_TS_INGEST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct atsc_ts_ingest atsc_ts_ingest;

typedef struct atsc_ts_ingest_stats {
    uint64_t bytes_in;        /* bytes pushed */
    uint64_t bytes_dropped;   /* bytes discarded while hunting for sync (including after a sync loss) */
    uint64_t packets_out;     /* packets popped, stuffed null packets included */
    uint64_t nulls_stuffed;   /* null packets inserted by pop_padded */
    uint64_t sync_errors;     /* packets passed through in lock although their sync byte was not 0x47 */
    uint64_t sync_losses;     /* times lock was lost */
    int locked;               /* 1 while in lock */
} atsc_ts_ingest_stats;

/* lock_packets: consecutive 0x47 at 188-byte spacing needed to declare lock (>=2; 0 -> 5, the usual MPEG-TS rule);
 * loss_packets: consecutive bad sync bytes that drop lock (>=1; 0 -> 3).  NULL on bad arguments / out of memory. */
atsc_ts_ingest* atsc_ts_ingest_create(int lock_packets, int loss_packets);
void atsc_ts_ingest_destroy(atsc_ts_ingest* h);

/* Append `n` bytes of the raw stream (any chunking).  Returns the number of aligned packets ready to pop. */
size_t atsc_ts_ingest_push(atsc_ts_ingest* h, const uint8_t* data, size_t n);

/* Aligned 188-byte packets ready to pop. */
size_t atsc_ts_ingest_available(const atsc_ts_ingest* h);

/* Pop up to max_packets packets into out (188 bytes each).  Returns the number written.  A packet passed through
 * during lock with a bad sync byte has that byte rewritten to 0x47 (counted in sync_errors). */
size_t atsc_ts_ingest_pop(atsc_ts_ingest* h, uint8_t* out, size_t max_packets);

/* Pop exactly `packets` packets: what the source has, then null packets (rate stuffing at a deadline / end-of-stream
 * padding to the 312-packet field the encoder consumes, reference atsc.cpp:72-75).  Returns the number of null packets
 * stuffed. */
size_t atsc_ts_ingest_pop_padded(atsc_ts_ingest* h, uint8_t* out, size_t packets);

void atsc_ts_ingest_get_stats(const atsc_ts_ingest* h, atsc_ts_ingest_stats* out);

/* Writes one null packet (188 bytes). */
void atsc_ts_null_packet(uint8_t* out188);

#ifdef __cplusplus
}
#endif
#endif /* ATSC_TS_INGEST_H */
