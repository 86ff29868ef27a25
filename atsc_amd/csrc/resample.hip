// SDR-rate polyphase resampler (include/atsc_hip_resample.h; SURVEY.md 8f-1 -- no reference counterpart, the
// reference runs the radio at the symbol rate: util/atsc_encoder.cpp:70).  HBM-bound: 8 B in, 8*L/M B (or less for
// the fixed-point formats) out per input sample; K complex FMAs per output.
//
// A block produces kOutPerBlock consecutive outputs of one channel: the input window they need (at most
// 2*kOutPerBlock + K samples for M <= 2L) is staged in LDS with coalesced 8-byte loads, the polyphase taps
// hp[phase][k] sit in LDS too (row stride K+1 floats: lanes walk phases with stride M mod L, the odd stride spreads
// them over the banks).  History before the call (K-1 samples per channel) comes from a small state buffer that a
// second tiny kernel refreshes after the main kernel has read it.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "atsc_hip.h"
#include "atsc_hip_resample.h"
#include "iq_format.h"

namespace atsc_k {
namespace {

constexpr int kOutPerBlock = 1024;
constexpr int kResThreads = 256;

template <int FMT>
__global__ __launch_bounds__(kResThreads) void k_resample(const float2* __restrict__ in, size_t n_in, size_t in_stride,
                                                          const float2* __restrict__ hist,
                                                          typename OutSample<FMT>::type* __restrict__ out, size_t n_out,
                                                          size_t out_stride, const float* __restrict__ hp, int L, int M,
                                                          int K, long q0) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* s_h = reinterpret_cast<float*>(smem);                        // [L][K+1]
    float2* s_x = reinterpret_cast<float2*>(s_h + ((L * (K + 1) + 3) & ~3));  // input window
    const int ch = blockIdx.y;
    const size_t j0 = (size_t)blockIdx.x * kOutPerBlock;
    const size_t j1 = min(n_out, j0 + kOutPerBlock);
    for (int i = threadIdx.x; i < L * K; i += kResThreads) s_h[(i / K) * (K + 1) + i % K] = hp[i];
    // local input index of output j: floor((q0 + j*M) / L); window [lo - (K-1), hi]
    const long lo = (q0 + (long)j0 * M) / L - (K - 1);
    const long hi = (q0 + (long)(j1 - 1) * M) / L;
    const float2* in_c = in + (size_t)ch * in_stride;
    const float2* hist_c = hist + (size_t)ch * (K - 1);
    for (long i = lo + threadIdx.x; i <= hi; i += kResThreads) {
        float2 v;
        if (i >= 0) v = i < (long)n_in ? in_c[i] : make_float2(0.f, 0.f);
        else v = hist_c[i + (K - 1)];          // i >= -(K-1) because lo >= -(K-1)
        s_x[i - lo] = v;
    }
    __syncthreads();
    typename OutSample<FMT>::type* out_c = out + (size_t)ch * out_stride;
    for (size_t j = j0 + threadIdx.x; j < j1; j += kResThreads) {
        const long t = q0 + (long)j * M;
        const long i = t / L;
        const int ph = (int)(t - i * L);
        const float* h = s_h + ph * (K + 1);
        const float2* x = s_x + (i - lo);       // x[-k] = input i-k
        float re = 0.f, im = 0.f;
        for (int k = 0; k < K; ++k) {
            const float c = h[k];
            const float2 v = x[-k];
            re = __fmaf_rn(c, v.x, re);
            im = __fmaf_rn(c, v.y, im);
        }
        out_c[j] = to_out<FMT>((v2f_t){re, im});
    }
}


// ---- integer upsampling fast path (M == 1, L in {2,3,4}, K == 16: the "2x the symbol rate" SDR case).
// The generic kernel reads K taps + K samples from LDS per output (192 B at K = 16) and is LDS-bandwidth-bound.
// Here a thread owns R = 16/L consecutive input positions and all L phases of each: its K+R-1 samples live in
// registers as a sliding window (one LDS read per new sample), the L*K taps are wave-uniform (scalar loads), and
// the 16 outputs of a thread leave through a padded LDS image so that the global stores are coalesced:
// ~28 B of LDS traffic per output instead of 192.  Phase 0 of the Nyquist prototype is the identity (centre tap
// 1.0, all others exactly 0: atsc_hip_resampler_design), so it is copied, not computed.
template <int L>
struct UpGeom {
    static constexpr int R = 16 / L;                 // input positions per thread
    static constexpr int kIn = kResThreads * R;      // input positions per block
    static constexpr int kOut = kIn * L;             // outputs per block
};

template <int L, int K, int FMT>
__global__ __launch_bounds__(kResThreads) void k_upsample(const float2* __restrict__ in, size_t n_in, size_t in_stride,
                                                          const float2* __restrict__ hist,
                                                          typename OutSample<FMT>::type* __restrict__ out,
                                                          size_t out_stride, const float* __restrict__ hp) {
    typedef UpGeom<L> G;
    constexpr int R = G::R;
    constexpr int kWin = G::kIn + K - 1;
    // sample w of the window sits at w + w / R (one pad per R: a thread's first read is 8*(R+1) bytes from its neighbour's)
    __shared__ float2 s_x[kWin + kWin / R + 1];
    __shared__ float2 s_y[G::kOut + G::kOut / 16 + 1];   // output o at o + o / 16
    const int tid = threadIdx.x;
    const int ch = blockIdx.y;
    const long i0 = (long)blockIdx.x * G::kIn;           // first input position of the block
    const float2* in_c = in + (size_t)ch * in_stride;
    const float2* hist_c = hist + (size_t)ch * (K - 1);
    for (int w = tid; w < kWin; w += kResThreads) {
        const long i = i0 - (K - 1) + w;
        float2 v;
        if (i >= 0) v = i < (long)n_in ? in_c[i] : make_float2(0.f, 0.f);
        else v = hist_c[i + (K - 1)];
        s_x[w + w / R] = v;
    }
    __syncthreads();
    // window of this thread: samples i0 + tid*R - (K-1) ... i0 + tid*R + R-1  = window slots tid*R ... tid*R + K+R-2
    v2f_t x[K + R - 1];
#pragma unroll
    for (int w = 0; w < K + R - 1; ++w) {
        const int ws = tid * R + w;
        const float2 v = s_x[ws + ws / R];
        x[w] = (v2f_t){v.x, v.y};
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int o = (tid * R + r) * L;
        // phase 0: y[L*i] = x[i - K/2]
        {
            const v2f_t y = x[r + K - 1 - K / 2];
            s_y[o + o / 16] = make_float2(y.x, y.y);
        }
#pragma unroll
        for (int ph = 1; ph < L; ++ph) {
            v2f_t acc = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float c = hp[ph * K + k];          // uniform: scalar load
                acc = __builtin_elementwise_fma(x[r + K - 1 - k], (v2f_t){c, c}, acc);
            }
            s_y[o + ph + (o + ph) / 16] = make_float2(acc.x, acc.y);
        }
    }
    __syncthreads();
    const size_t j0 = (size_t)i0 * L;
    const size_t n_out = n_in * L;
    typename OutSample<FMT>::type* out_c = out + (size_t)ch * out_stride + j0;
    const int n_here = (int)min((size_t)G::kOut, n_out - j0);
    for (int o = tid; o < n_here; o += kResThreads) {
        const float2 v = s_y[o + o / 16];
        out_c[o] = to_out<FMT>((v2f_t){v.x, v.y});
    }
}

// new history = last K-1 samples of [old history | this call's input]
__global__ void k_resample_history(const float2* __restrict__ in, size_t n_in, size_t in_stride, const float2* __restrict__ hist,
                                   float2* __restrict__ hist_new, int K) {
    const int ch = blockIdx.x;
    for (int i = threadIdx.x; i < K - 1; i += blockDim.x) {
        const long src = (long)n_in - (K - 1) + i;
        hist_new[(size_t)ch * (K - 1) + i] =
            src >= 0 ? in[(size_t)ch * in_stride + src] : hist[(size_t)ch * (K - 1) + (src + (K - 1))];
    }
}

double bessel_i0(double x) {
    double s = 1.0, t = 1.0;
    for (int k = 1; k < 64; ++k) {
        t *= (x / (2.0 * k)) * (x / (2.0 * k));
        s += t;
        if (t < 1e-18 * s) break;
    }
    return s;
}

}  // namespace
}  // namespace atsc_k

using namespace atsc_k;

struct atsc_hip_resampler {
    int device = 0, channels = 0, L = 1, M = 1, K = 2, fmt = 0;
    long q0 = 0;                  // m0*M - n0*L for the next output m0 / next input n0: in [0, max(L,M))
    float* d_hp = nullptr;        // [L][K] polyphase taps
    float2* d_hist[2] = {nullptr, nullptr};
    int cur = 0;
    bool nyquist = false;        // phase 0 is the identity with delay K/2 (integer-upsampling fast path)
};

static bool args_ok(int L, int M, int K) {
    return L >= 1 && M >= 1 && L <= ATSC_HIP_RESAMPLE_MAX_L && M <= 2 * L && K >= 2 && K <= ATSC_HIP_RESAMPLE_MAX_K && K % 2 == 0;
}

extern "C" int atsc_hip_resampler_design(int L, int M, int K, double beta, float* h_out) {
    if (!h_out || !args_ok(L, M, K)) return ATSC_HIP_ERR_ARG;
    if (beta <= 0) beta = 8.0;
    const int N = K * L, c = (K / 2) * L;
    const double fc = 1.0 / std::max(L, M);   // cutoff in units of pi at the interpolated rate
    const double i0b = bessel_i0(beta);
    for (int n = 0; n < N; ++n) {
        const double d = n - c;
        const double r = d / c;                                    // window support [-c, c]; n = 0 is its end point
        const double w = bessel_i0(beta * std::sqrt(std::max(0.0, 1.0 - r * r))) / i0b;
        const double a = M_PI * fc * d;
        const double sinc = d == 0 ? 1.0 : std::sin(a) / a;
        double v = L * fc * sinc * w;
        if (M <= L && (n - c) % L == 0) v = n == c ? 1.0 : 0.0;    // Nyquist property made exact
        h_out[n] = (float)v;
    }
    return ATSC_HIP_OK;
}

extern "C" void atsc_hip_resampler_destroy(atsc_hip_resampler* r) {
    if (!r) return;
    hipSetDevice(r->device);
    hipFree(r->d_hp);
    hipFree(r->d_hist[0]);
    hipFree(r->d_hist[1]);
    delete r;
}

extern "C" int atsc_hip_resampler_create(int device, int channels, int L, int M, int K, double beta, int out_format,
                                         atsc_hip_resampler** out) {
    if (!out || channels < 1 || !args_ok(L, M, K) || !atsc_hip_iq_sample_bytes(out_format)) return ATSC_HIP_ERR_ARG;
    *out = nullptr;
    if ((long)L * (K + 1) > 11000) return ATSC_HIP_ERR_TOO_LARGE;   // taps + input window stay below 64 KB of LDS
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ATSC_HIP_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return ATSC_HIP_ERR_ARG;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    auto* r = new (std::nothrow) atsc_hip_resampler();
    if (!r) return ATSC_HIP_ERR_ARG;
    r->device = device; r->channels = channels; r->L = L; r->M = M; r->K = K; r->fmt = out_format;
    std::vector<float> h((size_t)K * L), hp((size_t)K * L);
    atsc_hip_resampler_design(L, M, K, beta, h.data());
    for (int ph = 0; ph < L; ++ph)
        for (int k = 0; k < K; ++k) hp[(size_t)ph * K + k] = h[(size_t)k * L + ph];
    r->nyquist = true;
    for (int k = 0; k < K; ++k) r->nyquist = r->nyquist && hp[k] == (k == K / 2 ? 1.0f : 0.0f);
    const size_t hist_bytes = sizeof(float2) * (size_t)channels * (K - 1);
    if ((e = hipMalloc(&r->d_hp, sizeof(float) * hp.size())) != hipSuccess ||
        (e = hipMalloc(&r->d_hist[0], hist_bytes)) != hipSuccess || (e = hipMalloc(&r->d_hist[1], hist_bytes)) != hipSuccess ||
        (e = hipMemcpy(r->d_hp, hp.data(), sizeof(float) * hp.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemset(r->d_hist[0], 0, hist_bytes)) != hipSuccess) {
        atsc_hip_resampler_destroy(r);
        return (int)e;
    }
    *out = r;
    return ATSC_HIP_OK;
}

extern "C" size_t atsc_hip_resampler_out_count(const atsc_hip_resampler* r, size_t n_in) {
    if (!r) return 0;
    const long num = (long)n_in * r->L - r->q0;   // outputs j with q0 + j*M < n_in*L
    return num <= 0 ? 0 : (size_t)((num + r->M - 1) / r->M);
}

extern "C" int atsc_hip_resampler_reset(atsc_hip_resampler* r) {
    if (!r) return ATSC_HIP_ERR_ARG;
    hipError_t e = hipSetDevice(r->device);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemset(r->d_hist[r->cur], 0, sizeof(float2) * (size_t)r->channels * (r->K - 1));
    r->q0 = 0;
    return (int)e;
}

template <int FMT>
static void launch_resample(atsc_hip_resampler* r, hipStream_t s, const float2* in, size_t n_in, size_t in_stride, void* out,
                            size_t n_out, size_t out_stride) {
    typedef typename OutSample<FMT>::type out_t;
    const size_t window = 2 * (size_t)kOutPerBlock + r->K + 2;
    const size_t shmem = sizeof(float) * (((size_t)r->L * (r->K + 1) + 3) & ~(size_t)3) + sizeof(float2) * window;
    const dim3 grid((unsigned)((n_out + kOutPerBlock - 1) / kOutPerBlock), (unsigned)r->channels);
    hipLaunchKernelGGL(k_resample<FMT>, grid, dim3(kResThreads), shmem, s, in, n_in, in_stride, r->d_hist[r->cur],
                       static_cast<out_t*>(out), n_out, out_stride, r->d_hp, r->L, r->M, r->K, r->q0);
}

template <int L, int FMT>
static void launch_upsample(atsc_hip_resampler* r, hipStream_t s, const float2* in, size_t n_in, size_t in_stride, void* out,
                            size_t out_stride) {
    typedef typename OutSample<FMT>::type out_t;
    const dim3 grid((unsigned)((n_in + UpGeom<L>::kIn - 1) / UpGeom<L>::kIn), (unsigned)r->channels);
    hipLaunchKernelGGL((k_upsample<L, 16, FMT>), grid, dim3(kResThreads), 0, s, in, n_in, in_stride, r->d_hist[r->cur],
                       static_cast<out_t*>(out), out_stride, r->d_hp);
}

template <int FMT>
static bool try_upsample(atsc_hip_resampler* r, hipStream_t s, const float2* in, size_t n_in, size_t in_stride, void* out,
                         size_t out_stride) {
    if (r->M != 1 || r->K != 16 || !r->nyquist || r->q0 != 0) return false;
    switch (r->L) {
        case 2: launch_upsample<2, FMT>(r, s, in, n_in, in_stride, out, out_stride); return true;
        case 3: launch_upsample<3, FMT>(r, s, in, n_in, in_stride, out, out_stride); return true;
        case 4: launch_upsample<4, FMT>(r, s, in, n_in, in_stride, out, out_stride); return true;
        default: return false;
    }
}

extern "C" int atsc_hip_resample(atsc_hip_resampler* r, const void* d_in, size_t n_in, size_t in_stride, void* d_out,
                                 size_t out_stride, void* stream) {
    if (!r || !d_in || !d_out || n_in < 1 || in_stride < n_in) return ATSC_HIP_ERR_ARG;
    const size_t n_out = atsc_hip_resampler_out_count(r, n_in);
    if (out_stride < n_out) return ATSC_HIP_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(d_in) & 7) || (reinterpret_cast<uintptr_t>(d_out) & 7)) return ATSC_HIP_ERR_ARG;
    if (n_in > (size_t)1 << 40) return ATSC_HIP_ERR_TOO_LARGE;
    hipError_t e = hipSetDevice(r->device);
    if (e != hipSuccess) return (int)e;
    auto s = static_cast<hipStream_t>(stream);
    auto* in = static_cast<const float2*>(d_in);
    if (n_out) {
        switch (r->fmt) {
            case ATSC_HIP_IQ_CS16:
                if (!try_upsample<kIqCs16>(r, s, in, n_in, in_stride, d_out, out_stride))
                    launch_resample<kIqCs16>(r, s, in, n_in, in_stride, d_out, n_out, out_stride);
                break;
            case ATSC_HIP_IQ_CS8:
                if (!try_upsample<kIqCs8>(r, s, in, n_in, in_stride, d_out, out_stride))
                    launch_resample<kIqCs8>(r, s, in, n_in, in_stride, d_out, n_out, out_stride);
                break;
            default:
                if (!try_upsample<kIqCf32>(r, s, in, n_in, in_stride, d_out, out_stride))
                    launch_resample<kIqCf32>(r, s, in, n_in, in_stride, d_out, n_out, out_stride);
                break;
        }
    }
    hipLaunchKernelGGL(k_resample_history, dim3(r->channels), dim3(64), 0, s, in, n_in, in_stride, r->d_hist[r->cur],
                       r->d_hist[r->cur ^ 1], r->K);
    r->cur ^= 1;
    r->q0 += (long)n_out * r->M - (long)n_in * r->L;
    return (int)hipGetLastError();
}
