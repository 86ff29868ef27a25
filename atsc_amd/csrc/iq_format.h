// Device-side IQ sample formats shared by the FIR store (fir_fft.hip) and the resampler (resample.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.h"

namespace atsc_k {

typedef float v2f_t __attribute__((ext_vector_type(2)));

// ---- output sample formats (SDR sink adapter; the reference only ever hands out CF32,
// util/atsc_encoder.cpp:75).  CS16 / CS8 are the SoapySDR-style fixed-point formats, full scale 1.0 <-> 32767 / 127:
//   q = rint(clamp(y, -1, 1) * 32767)   (round half to even of the exact product; one v_cvt_pknorm_i16_f32)
//   q = rint(clamp(y, -1, 1) * 127)     (fma onto 1.5*2^23: the integer lands in the low mantissa bits)
template <int FMT> struct OutSample;
template <> struct OutSample<kIqCf32> { typedef float2 type; };
template <> struct OutSample<kIqCs16> { typedef uint32_t type; };
template <> struct OutSample<kIqCs8>  { typedef uint16_t type; };

template <int FMT>
__device__ __forceinline__ typename OutSample<FMT>::type to_out(v2f_t y) {
    if constexpr (FMT == kIqCf32) {
        return make_float2(y.x, y.y);
    } else if constexpr (FMT == kIqCs16) {
        uint32_t r;
        asm("v_cvt_pknorm_i16_f32 %0, %1, %2" : "=v"(r) : "v"(y.x), "v"(y.y));
        return r;
    } else {
        const float re = __builtin_amdgcn_fmed3f(y.x, -1.0f, 1.0f), im = __builtin_amdgcn_fmed3f(y.y, -1.0f, 1.0f);
        const float mx = __fmaf_rn(re, 127.0f, 12582912.0f), my = __fmaf_rn(im, 127.0f, 12582912.0f);
        // bytes: low byte of re's bits, low byte of im's bits
        return (uint16_t)__builtin_amdgcn_perm(__float_as_uint(my), __float_as_uint(mx), 0x0c0c0400u);
    }
}

}  // namespace atsc_k
