// Device bodies shared by the stand-alone kernels (k_trellis_scan in kernels.hip, k_osc_table in fir_fft.hip) and the
// merged launch of the fused path (k_scan_osc_state in kernels.hip): the carry scan of a batch uses 64 waves and waits for
// memory most of its 13 us, the oscillator table is a 134 MB write stream of 26 us that needs nothing from the TS, and the
// end-of-batch copy of the last interleaved bytes is a launch of its own for 192 lanes -- one grid with three kinds of
// workgroup runs them side by side.
#pragma once
#include "kernels.h"

namespace atsc_k {

__device__ __forceinline__ uint32_t wave_xor_inclusive(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v ^= o;
    }
    return v;
}

// One wave per channel: exclusive XOR scan over fields*26 parts, seeded with the channel's carry.
// Lane l owns one field (26 parts) per pass of 64 fields.
__device__ __forceinline__ void trellis_scan_body(const uint4* __restrict__ par, uint4* __restrict__ cin, uint4* __restrict__ carry,
                                                  uint4* __restrict__ span_par, const uint4* __restrict__ carry_xor, int fields,
                                                  int c, int lane) {
    uint4 init = carry[c];
    if (carry_xor) {  // time-split hand-off: state_in = own history XOR the parities of all earlier spans
        const uint4 x = carry_xor[c];
        init.x ^= x.x; init.y ^= x.y; init.z ^= x.z;
    }
    uint32_t run[3] = {init.x, init.y, init.z};
    for (int f0 = 0; f0 < fields; f0 += 64) {
        const int f = f0 + lane;
        const bool on = f < fields;
        const uint4* pp = par + ((size_t)c * fields + (on ? f : 0)) * kParts;
        uint32_t tot[3] = {0, 0, 0};
        if (on) {
            for (int p = 0; p < kParts; ++p) {
                const uint4 v = pp[p];
                tot[0] ^= v.x; tot[1] ^= v.y; tot[2] ^= v.z;
            }
        }
        uint32_t pre[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t inc = wave_xor_inclusive(tot[k], lane);
            pre[k] = run[k] ^ inc ^ tot[k];  // exclusive prefix for this lane's field
            run[k] ^= __shfl(inc, 63, 64);   // all lanes: total of this pass
        }
        if (on) {
            uint4* co = cin + ((size_t)c * fields + f) * kParts;
            for (int p = 0; p < kParts; ++p) {
                co[p] = make_uint4(pre[0], pre[1], pre[2], 0);
                const uint4 v = pp[p];
                pre[0] ^= v.x; pre[1] ^= v.y; pre[2] ^= v.z;
            }
        }
    }
    if (lane == 0) {
        carry[c] = make_uint4(run[0], run[1], run[2], 0);
        span_par[c] = make_uint4(run[0] ^ init.x, run[1] ^ init.y, run[2] ^ init.z, 0);
    }
}

// Oscillator table of a batch: tab[f] = T_(f0+f) for f < fields, out = T_(f0+fields)   (offset.h:35-41).
// planes != 0: the layout of the matrix-core forms (fir_fused.hip): [f][re plane | im plane] of kPaddedSyms dwords, where
// the dwords 2m, 2m+1 of a plane hold the positions 2m and 2m+1 as fp16 pieces of T * S (fp16's normal range; S = 4096 for
// fir_fused.hip, kOscPlanes4096, and 16 for fir_w.hip, kOscPlanes16):
// {Th(2m), Th(2m+1)}, {Tl(2m), Tl(2m+1)} with Th = fl16(T * S), Tl = fl16(T * S - Th).
// A thread owns two neighbouring table entries (pair m): the plane form packs their fp16 pieces into whole dwords (dword 2m
// of a plane = the two high pieces, dword 2m + 1 the two low ones), so it leaves as one 8-byte store per component and field
// instead of four 2-byte ones per entry; the complex form as one 16-byte store.
__device__ __forceinline__ void osc_table_body(const float2* __restrict__ in, float2* __restrict__ tab, float2* __restrict__ out,
                                               float2 sc, int fields, int planes, int m) {
    static_assert(kPaddedSyms % 2 == 0, "entries come in pairs");
    if (m >= kPaddedSyms / 2) return;
    const float4 t01 = reinterpret_cast<const float4*>(in)[m];
    float2 t[2] = {make_float2(t01.x, t01.y), make_float2(t01.z, t01.w)};
    float* pl = reinterpret_cast<float*>(tab);
    const float pscale = planes == kOscPlanes16 ? 16.0f : 4096.0f;
    for (int f = 0; f < fields; ++f) {
        if (planes) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                _Float16 hi[2], lo[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float v = (c ? t[j].y : t[j].x) * pscale;
                    hi[j] = (_Float16)v;
                    lo[j] = (_Float16)(v - (float)hi[j]);
                }
                uint2 w;
                w.x = (uint32_t)__builtin_bit_cast(uint16_t, hi[0]) | ((uint32_t)__builtin_bit_cast(uint16_t, hi[1]) << 16);
                w.y = (uint32_t)__builtin_bit_cast(uint16_t, lo[0]) | ((uint32_t)__builtin_bit_cast(uint16_t, lo[1]) << 16);
                *reinterpret_cast<uint2*>(pl + ((size_t)f * 2 + c) * kPaddedSyms + 2 * m) = w;
            }
        } else {
            reinterpret_cast<float4*>(tab + (size_t)f * kPaddedSyms)[m] = make_float4(t[0].x, t[0].y, t[1].x, t[1].y);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float t2 = __fmul_rn(t[j].y, sc.y);
            const float t3 = __fmul_rn(t[j].x, sc.y);
            t[j] = make_float2(__fmaf_rn(t[j].x, sc.x, -t2), __fmaf_rn(t[j].y, sc.x, t3));
        }
    }
    reinterpret_cast<float4*>(out)[m] = make_float4(t[0].x, t[0].y, t[1].x, t[1].y);
}

// (channel, dword) i of the last 12 interleaved bytes of the batch -> the history the next batch's first field starts from
__device__ __forceinline__ void save_last_group_body(const uint8_t* __restrict__ il, size_t il_stride, uint8_t* __restrict__ dst,
                                                     int channels, int fields, int i) {
    const int c = i / 3, w = i - 3 * c;
    if (c >= channels) return;
    const uint32_t* lg = reinterpret_cast<const uint32_t*>(il + (size_t)c * il_stride + (size_t)fields * kFieldCoded - 12);
    reinterpret_cast<uint32_t*>(dst + (size_t)c * 16)[w] = lg[w];
}

}  // namespace atsc_k
