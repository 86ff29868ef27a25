/*
 * atsc_hip_resample.h -- SDR-rate polyphase resampler behind the 8VSB modulator.
 *
 * The reference emits one sample per symbol and sets the radio itself to 10.762238 MS/s
 * (util/atsc_encoder.cpp:70, filter.h:9-14); radios whose clocks cannot make that rate need the stream at L/M times
 * it.  This stage has NO reference counterpart and no reference oracle: PARITY UNPINNED.  It is specified here and
 * checked (tests/test_gpu_resample.py) against a float64 model of the same formula, by the exact pass-through of
 * phase 0 for integer L (Nyquist prototype), by chunking invariance and by the image rejection measured on real
 * modulator output.
 *
 *   y[m] = sum_{k<K} h[k*L + (m*M mod L)] * x[floor(m*M/L) - k]          (x[<0] = 0 at stream start)
 *   h    = L * Kaiser(beta)-windowed sinc, cutoff pi/max(L,M), length K*L, centre (K/2)*L (K even); for M <= L the
 *          centre tap is exactly L... scaled to 1.0 and the taps at centre +- j*L are exactly 0, so for M == 1
 *          y[L*i] == x[i - K/2] bit for bit.
 * Group delay: K/2 input samples.  Streams continue across calls (K-1 samples of history and the phase are kept per
 * resampler; all channels advance in lockstep, like the encoder handle).
 */
#ifndef ATSC_HIP_RESAMPLE_H
#define ATSC_HIP_RESAMPLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct atsc_hip_resampler atsc_hip_resampler;

#define ATSC_HIP_RESAMPLE_MAX_L 256
#define ATSC_HIP_RESAMPLE_MAX_K 64

/* Prototype filter as the library designs it (host only, no GPU needed): h[0 .. K*L).  beta <= 0 -> 8.0 (~80 dB).
 * Returns 0, or ATSC_HIP_ERR_ARG for L,M < 1, L > MAX_L, M > 2L, K odd / < 2 / > MAX_K.
 * create additionally returns ATSC_HIP_ERR_TOO_LARGE when the L x (K+1) tap table does not fit the kernel's LDS budget
 * (L*(K+1) <= 11000). */
int atsc_hip_resampler_design(int interp_L, int decim_M, int taps_per_phase_K, double beta, float* h_out);

/* out_format: ATSC_HIP_IQ_CF32 / _CS16 / _CS8 (atsc_hip.h).  Input is always complex64. */
int atsc_hip_resampler_create(int device, int channels, int interp_L, int decim_M, int taps_per_phase_K, double beta,
                              int out_format, atsc_hip_resampler** out);
void atsc_hip_resampler_destroy(atsc_hip_resampler* r);

/* Number of output samples per channel the next atsc_hip_resample call with n_in input samples will write. */
size_t atsc_hip_resampler_out_count(const atsc_hip_resampler* r, size_t n_in);

/* d_in : [channels] rows of n_in complex64, in_stride samples apart (e.g. the d_iq of atsc_hip_encode_batch with
 *        in_stride = fields*260416);  d_out: [channels] rows of out_count samples of out_format, out_stride apart.
 * Asynchronous on `stream` (hipStream_t as void*).  Returns 0 / ATSC_HIP_ERR_* / a hipError_t. */
int atsc_hip_resample(atsc_hip_resampler* r, const void* d_in, size_t n_in, size_t in_stride, void* d_out,
                      size_t out_stride, void* stream);

/* Back to the start of a stream: zero history, phase 0. */
int atsc_hip_resampler_reset(atsc_hip_resampler* r);

#ifdef __cplusplus
}
#endif
#endif /* ATSC_HIP_RESAMPLE_H */
