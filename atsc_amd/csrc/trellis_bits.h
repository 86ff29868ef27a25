// Bit-sliced precoder / trellis helpers shared by k_interleave, k_trellis_map (kernels.hip) and the fused FIR
// (fir_fused.hip).  reference: trellis.h:10-51,121-499 (restated).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.h"

namespace atsc_k {

// A byte-time g (12 interleaved bytes) feeds one byte to each of the 12 encoders.  Encoder e takes
// IL[12g + ((e - 4*sigma) mod 12)], sigma = floor(12g/207) mod 3, so the three
// dwords of a group are whole 4-encoder words that only rotate: E_k = D_((k - sigma) mod 3) holds the
// bytes of encoders 4k..4k+3.  All per-encoder work is done on 4 encoders at once inside a dword.
//
// Per byte (4 symbols, MSB pair first; x2 = odd bits, x1 = even bits):
//   Z2[tau] = c2 ^ x2[0..tau]                       (precoder, inclusive prefix)
//   Z0[0]=sA, Z0[1]=sB^x1[0], Z0[2]=sA^x1[1], Z0[3]=sB^x1[0]^x1[2]   (Z0[t] = x1[t-1] ^ Z0[t-2])
// with the carries (c2, sA, sB) entering the byte; after it c2 ^= parity(x2), sA ^= x1[1]^x1[3],
// sB ^= x1[0]^x1[2].  Carries/parities are packed per encoder byte as bit0=c2, bit1=sA, bit2=sB.
struct Words3 { uint32_t w[3]; };

__device__ __forceinline__ int sigma_of(int g) { return ((12 * g) / kSegCoded) % 3; }

__device__ __forceinline__ Words3 canonical_group(const uint32_t* p, int g) {
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    const int sg = sigma_of(g);
    Words3 e;
    e.w[0] = sg == 0 ? d0 : (sg == 1 ? d2 : d1);
    e.w[1] = sg == 0 ? d1 : (sg == 1 ? d0 : d2);
    e.w[2] = sg == 0 ? d2 : (sg == 1 ? d1 : d0);
    return e;
}

__device__ __forceinline__ void trellis_local(uint32_t e, uint32_t& z2, uint32_t& z0, uint32_t& par) {
    uint32_t y = e & 0xAAAAAAAAu;
    y ^= (y >> 2) & 0x2A2A2A2Au;
    y ^= (y >> 4) & 0x0A0A0A0Au;
    z2 = y;
    const uint32_t u = (e >> 2) & 0x15151515u;
    z0 = u ^ ((u >> 4) & 0x01010101u);
    const uint32_t x1 = e & 0x55555555u;
    const uint32_t p2 = (y >> 1) & 0x01010101u;
    const uint32_t pa = ((x1 >> 4) ^ x1) & 0x01010101u;
    const uint32_t pb = ((x1 >> 6) ^ (x1 >> 2)) & 0x01010101u;
    par = p2 | (pa << 1) | (pb << 2);
}


}  // namespace atsc_k
