// gfx950 mixer + RRC FIR as overlap-save fast convolution.
//
//   y[i] = sum_{k<111} c[k] * x[i+k] * T_f[i+k]          (reference: offset.h:11-45, filter.h:9-69)
//
// The direct form costs 111 packed FMAs per output and is fp32-VALU-bound at ~0.2 of the HBM roof.  Here
// the field is cut into 1024-sample blocks that advance by 896 (overlap 128 >= 110); each block is
// transformed (1024-point complex FFT), multiplied by the conjugate tap spectrum and transformed back:
// ~38 packed VALU instructions per output.  fp32 round-off stays at 2.4e-7 of full scale (the direct
// in-order sum: 5.4e-7; tolerance 1e-5).
//
// Mapping: 1024 = 32 x 32.  One FFT block is owned by 32 lanes (half a wave); every lane keeps 32 complex
// points in registers and runs four 32-point in-register FFTs per block (two forward, two inverse), with
// ONE transposition through LDS per 1024-point transform:
//   forward  pass 1: lane n1 transforms x[n1 + 32*n2] over n2 -> kA; twiddle W^(n1*kA); LDS transpose
//            pass 2: lane kA transforms over n1 -> X[kA + 32*kB]
//   multiply by Hs[kA + 32*kB] = conj(FFT(c))/1024
//   inverse  pass 1: lane kA transforms over kB -> nB; twiddle W^-(nB*kA); LDS transpose
//            pass 2: lane nB transforms over kA -> y[nB + 32*nA]
// so both the symbol loads (32 consecutive bytes per lane, re-spread through LDS) and the IQ stores (32
// lanes x 8 B = 256 contiguous, 128-byte aligned bytes per output row) are coalesced and no bit-reversal
// pass exists: every 32-point transform is decimation-in-time with its input registers *named* in
// bit-reversed order.  Half-waves never synchronise with each other: LDS transposes are private to the
// 32 lanes of a block and ordered by the wave's in-order LDS queue.
//
// The oscillator T_f is read from a per-field table built once per batch by k_osc_table with the
// reference's exact fp32 recursion (offset.h:35-41), so a task (block, field, channel group) needs no
// history and the grid is a flat list of equal tasks.
#include "kernels.h"
#include "front_roles.h"
#include "iq_format.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace atsc_k {

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kRowPad = 34;                       // LDS row stride (complex) of a 32x32 transpose: 16-byte aligned rows, column
                                                  // writes (b64) and row reads (b128) both conflict-free
constexpr int kRowsLast = (kFieldSyms - (kFftBlocks - 1) * kFftValid) / 32;  // 18
constexpr int kTwAhead = 3;                       // twiddle pairs (16 B each) read ahead of use
constexpr int kHalfWaves = 16;                    // FFT blocks per workgroup step (512 threads)
constexpr int kFftThreads = 32 * kHalfWaves;

__device__ float4 g_fft_tab[1024];                // [0,512): TW pairs [a/2][b] = {W^(a*b), W^((a+1)*b)};  [512,1024): Hs pairs [kB/2][kA]

__host__ __device__ constexpr int br5(int p) {
    return ((p & 1) << 4) | ((p & 2) << 2) | (p & 4) | ((p & 8) >> 2) | ((p & 16) >> 4);
}

// cos / sin of 2*pi*m/32, m = 0..15
__device__ constexpr float kC32[16] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                                       0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                                       0.19509032201612826785f, 0.0f, -0.19509032201612826785f, -0.38268343236508977173f,
                                       -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                                       -0.92387953251128675613f, -0.98078528040323044913f};
__device__ constexpr float kS32[16] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                                       0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f,
                                       0.98078528040323044913f, 1.0f, 0.98078528040323044913f, 0.92387953251128675613f,
                                       0.83146961230254523708f, 0.70710678118654752440f, 0.55557023301960222474f,
                                       0.38268343236508977173f, 0.19509032201612826785f};

__device__ __forceinline__ v2f swap(v2f v) { return __builtin_shufflevector(v, v, 1, 0); }
__device__ __forceinline__ v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// a * w and a * conj(w), w = (wr, wi) a run-time value: 2 packed instructions, swap and sign in the modifiers
__device__ __forceinline__ v2f cmul(v2f a, v2f w) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
__device__ __forceinline__ v2f cmul_conj(v2f a, v2f w) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}

// (a.x + b.y, a.y - b.x) and (a.x - b.y, a.y + b.x): a +- (b rotated by -+90 degrees) in one instruction
__device__ __forceinline__ v2f add_swap_neghi(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f add_swap_neglo(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// One decimation-in-time butterfly (a, b) -> (a + w b, a - w b), w = W_32^(+-m) (forward: exp(-j..)).
template <bool INV>
__device__ __forceinline__ void butterfly(v2f& a, v2f& b, const int m) {
    constexpr float r = 0.70710678118654752440f;
    if (m == 0) {
        const v2f t = a;
        a = t + b;
        b = t - b;
    } else if (m == 8) {  // w = -j (forward), +j (inverse): b*w = (b.y, -b.x) / (-b.y, b.x)
        const v2f t = a;
        if (INV) { a = add_swap_neglo(t, b); b = add_swap_neghi(t, b); }
        else     { a = add_swap_neghi(t, b); b = add_swap_neglo(t, b); }
    } else if (m == 4 || m == 12) {
        // forward m=4: r(1-j)   m=12: -r(1+j);   inverse m=4: r(1+j)   m=12: -r(1-j)
        const bool minus_j = (m == 4) != INV;
        const v2f s = minus_j ? add_swap_neghi(b, b) : add_swap_neglo(b, b);  // (bx+by, by-bx) : (bx-by, by+bx)
        const float g = (m == 4) ? r : -r;
        const v2f t = a;
        a = vfma(s, (v2f){g, g}, t);
        b = vfma(s, (v2f){-g, -g}, t);
    } else {
        const float wr = kC32[m];
        const float wi = INV ? kS32[m] : -kS32[m];
        const v2f u = vfma(b, (v2f){wr, wr}, a);
        const v2f p = vfma(swap(b), (v2f){-wi, wi}, u);
        b = vfma(a, (v2f){2.0f, 2.0f}, -p);
        a = p;
    }
}

// 32-point DIT FFT in registers: x[p] holds input element br5(p); on return x[k] holds output element k.
// `done(k)` is called as soon as outputs k and k+16 are final, so a caller can stream them out.
template <bool INV, typename F>
__device__ __forceinline__ void fft32(v2f (&x)[32], F done) {
#pragma unroll
    for (int half = 1; half < 16; half <<= 1) {
#pragma unroll
        for (int g = 0; g < 32; g += 2 * half) {
#pragma unroll
            for (int i = 0; i < half; ++i) butterfly<INV>(x[g + i], x[g + i + half], i * (16 / half));
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        butterfly<INV>(x[i], x[i + 16], i);
        done(i);
    }
}
template <bool INV>
__device__ __forceinline__ void fft32(v2f (&x)[32]) {
    fft32<INV>(x, [](int) {});
}

__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave execute in order; this only stops the compiler from moving them.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float level_of_byte(uint32_t v) {
    return __fmaf_rn((float)v, 2.0f, -5.75f);  // 2v - 7 + 1.25 (signal.h:16), exact
}

// 32 consecutive symbols of the padded field starting at pos (multiple of 32): the field itself, then the copy of
// its sync head (field_sync.h:16).  Positions >= 260416 + 820 are never reached (host-checked geometry).
__device__ __forceinline__ void load_syms32(const uint8_t* sym_f, int pos, uint4& lo, uint4& hi) {
    const int src = pos < kFieldSyms ? pos : pos - kFieldSyms;
    const uint4* p = reinterpret_cast<const uint4*>(sym_f + src);
    lo = p[0];
    hi = p[1];
}

// A lane's row of 32 transposed values, 16 bytes at a time, into bit-reversed register order:
// elements 2q, 2q+1 -> registers br5(2q), br5(2q)+16.
__device__ __forceinline__ void load_row(v2f (&x)[32], const float4* row) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float4 v = row[q];
        x[br5(2 * q)] = (v2f){v.x, v.y};
        x[br5(2 * q) + 16] = (v2f){v.z, v.w};
    }
}

#define TABREAD(e) (e)
#define LOADROW(x, row) load_row(x, row)
#define STOREEL(x, k, v0, v1) do { *reinterpret_cast<v2f*>(&xs[(k) * kRowPad + t]) = v0; *reinterpret_cast<v2f*>(&xs[((k) + 1) * kRowPad + t]) = v1; } while (0)



// ---- phases of one 1024-point block (all on the 32 lanes that own it; `xs` is the block's LDS transpose buffer)

// symbols: 32 consecutive bytes per lane -> LDS
__device__ __forceinline__ void stage_syms(float2* xs, int t, const uint4& lo, const uint4& hi) {
    reinterpret_cast<uint4*>(xs)[2 * t] = lo;
    reinterpret_cast<uint4*>(xs)[2 * t + 1] = hi;
}
// byte t + 32*j per lane, level, times the oscillator: x * T as two single-rounded products (offset.h:24-31)
__device__ __forceinline__ void mix(v2f (&x)[32], const float2* xs, int t, const v2f (&o)[32]) {
    const uint8_t* xs8 = reinterpret_cast<const uint8_t*>(xs);
#pragma unroll
    for (int p = 0; p < 32; ++p) {
        const float lv = level_of_byte(xs8[t + 32 * br5(p)]);
        x[p] = o[p] * (v2f){lv, lv};
    }
}
// transform over the registers, twiddle W^(+-t*k), write column t of the transpose, then start reading row t back
template <bool INV>
__device__ __forceinline__ void pass1(v2f (&x)[32], float2* xs, int t, const float4* tw) {
    __builtin_amdgcn_sched_barrier(0);  // phases stay in program order: the other stream's transpose is in flight
    fft32<INV>(x);
    __builtin_amdgcn_sched_barrier(0);
    // twiddle table reads run kTwAhead pairs ahead of their use
    float4 wq[kTwAhead];
#pragma unroll
    for (int d = 0; d < kTwAhead; ++d) wq[d] = TABREAD(tw[32 * d]);
#pragma unroll
    for (int k = 0; k < 32; k += 2) {
        const float4 w = wq[(k / 2) % kTwAhead];
        if (k / 2 + kTwAhead < 16) wq[(k / 2) % kTwAhead] = TABREAD(tw[16 * (k + 2 * kTwAhead)]);
        const v2f w0 = (v2f){w.x, w.y}, w1 = (v2f){w.z, w.w};
        const v2f v0 = k == 0 ? x[0] : (INV ? cmul_conj(x[k], w0) : cmul(x[k], w0));
        const v2f v1 = INV ? cmul_conj(x[k + 1], w1) : cmul(x[k + 1], w1);
        STOREEL(x, k, v0, v1);
        __builtin_amdgcn_sched_barrier(0);
    }
    wave_lds_sync();
    LOADROW(x, reinterpret_cast<const float4*>(xs + t * kRowPad));
    wave_lds_sync();
    __builtin_amdgcn_sched_barrier(0);
}
// forward pass 2 (over n1) and the spectrum multiply; x[kB] = X[t + 32*kB] becomes the inverse's input register br5(kB)
__device__ __forceinline__ void pass2_spectrum(v2f (&x)[32], const float4* hs) {
    fft32<false>(x);
    v2f y[32];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float4 h = TABREAD(hs[32 * q]);
        y[br5(2 * q)] = cmul(x[2 * q], (v2f){h.x, h.y});
        y[br5(2 * q) + 16] = cmul(x[2 * q + 1], (v2f){h.z, h.w});
    }
#pragma unroll
    for (int p = 0; p < 32; ++p) x[p] = y[p];
}
// inverse pass 2 (over kA) -> y[t + 32*nA]; rows nA < 28 are free of circular wrap.  Rows leave as soon as the last
// butterfly stage has produced them.  dst_lo serves rows 0..17, dst_hi rows 18..27 (a trash area for the last block
// of a field, which has only 18 rows): always 28 stores, so every vmcnt wait in the loop is exact.
template <int FMT>
__device__ __forceinline__ void pass4_store(v2f (&y)[32], typename OutSample<FMT>::type* dst_lo,
                                            typename OutSample<FMT>::type* dst_hi) {
    fft32<true>(y, [&](int i) {
        dst_lo[32 * i] = to_out<FMT>(y[i]);
        if (i + 16 < kRowsLast) dst_lo[32 * (i + 16)] = to_out<FMT>(y[i + 16]);
        else if (i + 16 < kFftValid / 32) dst_hi[32 * (i + 16)] = to_out<FMT>(y[i + 16]);
    });
}

}  // namespace

// One task = (FFT block b, field f, channel group): a half-wave walks the group's channels two at a time.  The two
// streams A and B share the oscillator registers and the LDS buffer and are interleaved phase by phase, so that the
// LDS round trip of one stream's transpose and the drain of its IQ stores run under the other stream's butterflies
// (LDS operations of a wave execute in program order, which is all the ordering the shared buffer needs).
template <int FMT>
__global__ __launch_bounds__(kFftThreads) void k_mix_fir_fft(const uint8_t* __restrict__ sym, size_t sym_stride,
                                                             typename OutSample<FMT>::type* __restrict__ iq, size_t iq_stride,
                                                             const float2* __restrict__ osc_tab,
                                                             typename OutSample<FMT>::type* __restrict__ trash,
                                                             int channels, int fields, int ch_per_group, int ch_groups) {
    typedef typename OutSample<FMT>::type out_t;
    __shared__ float4 s_tw[512];
    __shared__ float4 s_h[512];
    __shared__ __align__(16) float2 s_x[kHalfWaves][32 * kRowPad];
    const int tid = threadIdx.x;
    const int t = tid & 31;
    const int hw = tid >> 5;

    s_tw[tid] = g_fft_tab[tid];
    s_h[tid] = g_fft_tab[512 + tid];
    __syncthreads();

    const long task = (long)blockIdx.x * kHalfWaves + hw;
    const int b = (int)(task % kFftBlocks);
    const long g = task / kFftBlocks;
    const int cg = (int)(g % ch_groups);
    const int f = (int)(g / ch_groups);
    if (f >= fields) return;
    const int c_lo = cg * ch_per_group, c_hi = min(channels, c_lo + ch_per_group);
    const int base = b * kFftValid;  // first sample / first output of the block

    // oscillator slice of the block, in the register order of the first transform: o[p] = T_f[base + t + 32*br5(p)]
    v2f o[32];
    {
        const float2* of = osc_tab + (size_t)f * kPaddedSyms + base + t;
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            const float2 v = of[32 * br5(p)];
            o[p] = (v2f){v.x, v.y};
        }
    }

    float2* xs = s_x[hw];
    const float4* tw = s_tw + t;
    const float4* hs = s_h + t;
    const uint8_t* sym_f = sym + (size_t)f * kFieldSyms;
    out_t* iq_f = iq + (size_t)f * kFieldSyms + base + t;
    out_t* trash_t = trash + t;
    const bool full_block = b != kFftBlocks - 1;

    // stream B of the last pair repeats stream A when the group has an odd number of channels; its rows go to trash
    uint4 a_lo, a_hi, b_lo, b_hi;
    load_syms32(sym_f + (size_t)c_lo * sym_stride, base + 32 * t, a_lo, a_hi);
    load_syms32(sym_f + (size_t)min(c_lo + 1, c_hi - 1) * sym_stride, base + 32 * t, b_lo, b_hi);

    // Every load issued so far is complete before the loop is entered: the in-order vmcnt wait for the symbol
    // prefetch at the top of an iteration then never has to cover the previous iteration's IQ stores.
#pragma unroll
    for (int p = 0; p < 32; ++p) asm volatile("" ::"v"(o[p]));
    asm volatile("" ::"v"(a_lo.x), "v"(a_lo.y), "v"(a_lo.z), "v"(a_lo.w), "v"(a_hi.x), "v"(a_hi.y), "v"(a_hi.z), "v"(a_hi.w));
    asm volatile("" ::"v"(b_lo.x), "v"(b_lo.y), "v"(b_lo.z), "v"(b_lo.w), "v"(b_hi.x), "v"(b_hi.y), "v"(b_hi.z), "v"(b_hi.w));

    for (int c = c_lo; c < c_hi; c += 2) {
        v2f xa[32], xb[32];
        // ---- symbols of both streams through LDS, next pair's symbols prefetched
        stage_syms(xs, t, a_lo, a_hi);
        stage_syms(xs + 128, t, b_lo, b_hi);   // 1 KB further on: one LDS round trip for both streams
        wave_lds_sync();
        mix(xa, xs, t, o);
        mix(xb, xs + 128, t, o);
        wave_lds_sync();
        {
            const int ca = min(c + 2, c_hi - 1), cb = min(c + 3, c_hi - 1);
            load_syms32(sym_f + (size_t)ca * sym_stride, base + 32 * t, a_lo, a_hi);
            load_syms32(sym_f + (size_t)cb * sym_stride, base + 32 * t, b_lo, b_hi);
        }
        // ---- forward: A's transpose is in flight while B's first pass runs, and so on
        pass1<false>(xa, xs, t, tw);
        pass1<false>(xb, xs, t, tw);
        pass2_spectrum(xa, hs);
        pass1<true>(xa, xs, t, tw);
        pass2_spectrum(xb, hs);
        pass1<true>(xb, xs, t, tw);
        // ---- inverse pass 2 and the stores
        const bool b_real = c + 1 < c_hi;
        out_t* da = iq_f + (size_t)c * iq_stride;
        out_t* db = b_real ? iq_f + (size_t)(c + 1) * iq_stride : trash_t;
        pass4_store<FMT>(xa, da, full_block ? da : trash_t);
        pass4_store<FMT>(xb, db, full_block ? db : trash_t);
    }
}

// Oscillator table of a batch (front_roles.h: osc_table_body).
__global__ __launch_bounds__(256) void k_osc_table(const float2* __restrict__ in, float2* __restrict__ tab,
                                                   float2* __restrict__ out, float2 sc, int fields, int planes) {
    osc_table_body(in, tab, out, sc, fields, planes, blockIdx.x * 256 + threadIdx.x);
}

hipError_t upload_fft_tables(const float* taps112) {
    static_assert(kFftBlocks * kFftValid >= kFieldSyms, "blocks must cover the field");
    static_assert((kFftBlocks - 1) * kFftValid + kFftN <= kFieldSyms + 820, "look-ahead must stay inside the sync-head copy");
    static_assert(kFftN - kFftValid >= 110, "overlap must cover the filter");
    static_assert((kFieldSyms - (kFftBlocks - 1) * kFftValid) % 32 == 0, "field end must fall on a 32-sample row");
    std::vector<float> h(4 * 1024);
    const double w0 = -2.0 * M_PI / kFftN;
    // TW pairs: entry [a/2][b] holds W^(a*b) (a even) then W^((a+1)*b)
    for (int a = 0; a < 32; ++a)
        for (int b = 0; b < 32; ++b) {
            float* e = &h[4 * (32 * (a / 2) + b) + 2 * (a & 1)];
            e[0] = (float)std::cos(w0 * a * b);
            e[1] = (float)std::sin(w0 * a * b);
        }
    // Hs[k] = conj(sum_n c[n] W^(n k)) / N with k = kA + 32 kB, stored as pairs [kB/2][kA]
    for (int k = 0; k < kFftN; ++k) {
        double re = 0, im = 0;
        for (int n = 0; n < 111; ++n) {
            re += (double)taps112[n] * std::cos(w0 * n * k);
            im += (double)taps112[n] * std::sin(w0 * n * k);
        }
        const int kA = k & 31, kB = k >> 5;
        float* e = &h[4 * (512 + 32 * (kB / 2) + kA) + 2 * (kB & 1)];
        e[0] = (float)(re / kFftN);
        e[1] = (float)(-im / kFftN);
    }
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fft_tab), h.data(), sizeof(float) * h.size());
}

void launch_osc_table(hipStream_t s, const float2* osc_in, float2* osc_out, float2* osc_tab, float2 scale, int fields,
                      int planes) {
    hipLaunchKernelGGL(k_osc_table, dim3((kPaddedSyms / 2 + 255) / 256), dim3(256), 0, s, osc_in, osc_tab, osc_out, scale,
                       fields, planes);
}

template <int FMT>
static void launch_fmt(hipStream_t s, unsigned grid, const uint8_t* d_sym, size_t sym_stride, void* d_iq, size_t iq_stride,
                       const float2* osc_tab, float2* trash, int channels, int fields, int ch_per_group, int ch_groups) {
    typedef typename OutSample<FMT>::type out_t;
    hipLaunchKernelGGL(k_mix_fir_fft<FMT>, dim3(grid), dim3(kFftThreads), 0, s, d_sym, sym_stride, static_cast<out_t*>(d_iq),
                       iq_stride, osc_tab, reinterpret_cast<out_t*>(trash), channels, fields, ch_per_group, ch_groups);
}

void launch_mix_fir_fft(hipStream_t s, const uint8_t* d_sym, size_t sym_stride, void* d_iq, size_t iq_stride,
                        float2* osc_tab, int channels, int fields, int iq_format) {
    // Channels per task: as many as possible (the oscillator slice of a task is read once and shared by its channels),
    // but at least ~8 rounds of tasks over the 4096 half-wave slots of the chip so that the flat task list balances.
    const long per_group = (long)kFftBlocks * fields;              // tasks per channel group
    const long groups = (8L * 4096 + per_group - 1) / per_group;   // groups wanted
    int ch_per_group = (int)(channels / (groups < 1 ? 1 : groups));
    ch_per_group &= ~1;                                            // whole pairs
    if (ch_per_group > 32) ch_per_group = 32;
    if (ch_per_group < 2) ch_per_group = 2;
    if (ch_per_group > channels) ch_per_group = channels;
    const int ch_groups = (channels + ch_per_group - 1) / ch_per_group;
    const long tasks = (long)kFftBlocks * ch_groups * fields;
    const unsigned grid = (unsigned)((tasks + kHalfWaves - 1) / kHalfWaves);
    float2* trash = osc_tab + (size_t)fields * kPaddedSyms;  // kOscTabSlack complex behind the table
    switch (iq_format) {
        case kIqCs16: launch_fmt<kIqCs16>(s, grid, d_sym, sym_stride, d_iq, iq_stride, osc_tab, trash, channels, fields, ch_per_group, ch_groups); break;
        case kIqCs8: launch_fmt<kIqCs8>(s, grid, d_sym, sym_stride, d_iq, iq_stride, osc_tab, trash, channels, fields, ch_per_group, ch_groups); break;
        default: launch_fmt<kIqCf32>(s, grid, d_sym, sym_stride, d_iq, iq_stride, osc_tab, trash, channels, fields, ch_per_group, ch_groups); break;
    }
}

}  // namespace atsc_k
