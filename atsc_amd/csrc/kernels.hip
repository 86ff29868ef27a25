// gfx950 (MI355X / CDNA4) kernels of the ATSC 8VSB modulator hot path.
//
//   k_rs_parity      RS(207,187) parity of the randomized packets (PRBS XOR by linearity)   reference: randomize.h:43-54, reed_solomon.h:192-242
//   k_interleave     52-branch convolutional byte interleaver as an LDS shear   reference: interleaver.h:11-108
//                    (+ the per-part GF(2) parities of the 12 precoders/trellis encoders: trellis.h:10-51,240-242)
//   k_trellis_scan   exclusive XOR scan of the parities -> carry entering every part
//   k_trellis_map    precoder + trellis + 8-level map + segment/field sync -> u8 symbols   reference: trellis.h:121-499,549-576, field_sync.h:12-26
//   (k_mix_fir_fft   pilot/offset mixer + 112-tap RRC FIR -> complex64: fir_fft.hip)
//   k_osc_advance    oscillator table advance by n fields                        reference: offset.h:35-41
//   k_save_state     end-of-batch history (interleaver tail, last group)
//
// All byte/bit stages are bit-exact restatements of the closed forms in SURVEY.md Appendix A; the float
// stage (fir_fft.hip) follows the reference's rounding for the oscillator recursion.
#include "kernels.h"
#include "trellis_bits.h"

#include <cstdlib>


namespace atsc_k {

// ======================================================================================= K1 rand + RS
// Two kernels.
//
// k_rs_parity: one lane = one TS packet, held whole in registers (47 dwords, read straight from HBM), and the
// systematic LFSR run four data bytes per step: with u = (first four state bytes) ^ (next four data bytes) the new
// state is shift4(state) ^ T4[u0] ^ T3[u1] ^ T2[u2] ^ T1[u3] (tables.cpp) -- per TS dword one XOR, four byte
// extracts, 8 LDS reads (four 20-byte rows) and 20 XORs; the 4-byte shift of the 160-bit register is a renaming.
// The dropped TS sync byte (randomize.h:48) is fed as a zero, which leaves a zero state untouched, so a packet is
// exactly 47 steps.  The code is GF(2)-linear, so the raw TS bytes are encoded and the (precomputed) parity of the
// packet's PRBS row is added at the end.  The row lookups are the cost: lanes read random rows, so the tables sit in
// LDS in a bank-spread image (4 copies of the 16-byte part, 8 of the 4-byte part; lanes that share a copy collide only
// within v mod 4), and nothing else lives in LDS so that 16 waves per CU hide the lookup latency.
//
// k_assemble (debug tap, ATSC_HIP_FLAG_TAPS): coded[207 s + k] = ts[188 s + 1 + k] ^ prbs[187 s + k] (k < 187) followed
// by the 20 parity bytes: a byte-shifting copy, staged through LDS rows of 212 bytes with coalesced loads and stores.
// The path itself never materialises the coded stream: k_interleave reads TS + parity directly.
constexpr int kRsThreads = 1024;

typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));

__global__ __launch_bounds__(kRsThreads) void k_rs_parity(const uint8_t* __restrict__ ts, const uint4* __restrict__ rs_lds,
                                                          const uint4* __restrict__ prbs_parity,
                                                          uint4* __restrict__ parity, long npackets) {
    __shared__ uint4 s_a[4 * 256 * 4];      // 65536 B
    __shared__ uint32_t s_e[4 * 256 * 8];   // 32768 B
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) s_a[tid + kRsThreads * i] = rs_lds[tid + kRsThreads * i];
#pragma unroll
    for (int i = 0; i < 2; ++i) reinterpret_cast<uint4*>(s_e)[tid + kRsThreads * i] = rs_lds[4096 + tid + kRsThreads * i];
    __syncthreads();

    const uint4* ta = s_a + (tid & 3);
    const uint32_t* te = s_e + (tid & 7);
    for (long P = (long)blockIdx.x * kRsThreads + tid; P < npackets; P += (long)gridDim.x * kRsThreads) {
        uint32_t d[48];
        {
            const u32x4a4* src = reinterpret_cast<const u32x4a4*>(ts + P * kTsBytes);
#pragma unroll
            for (int i = 0; i < 11; ++i) {
                const u32x4a4 v = src[i];
                d[4 * i] = v.x; d[4 * i + 1] = v.y; d[4 * i + 2] = v.z; d[4 * i + 3] = v.w;
            }
            const uint32_t* tail = reinterpret_cast<const uint32_t*>(ts + P * kTsBytes) + 44;
            d[44] = tail[0]; d[45] = tail[1]; d[46] = tail[2];
        }
        d[0] &= 0xffffff00u;  // TS sync byte: dropped unchecked (randomize.h:48)
        uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0;
#pragma unroll
        for (int w = 0; w < 47; ++w) {
            const uint32_t u = d[w] ^ b0;
            // first byte in time = lowest byte = four steps from the end -> table T4 (index 3)
            const uint32_t i4 = 3 * 256 + (u & 0xffu), i3 = 2 * 256 + ((u >> 8) & 0xffu), i2 = 256 + ((u >> 16) & 0xffu),
                           i1 = u >> 24;
            const uint4 r4 = ta[i4 * 4], r3 = ta[i3 * 4], r2 = ta[i2 * 4], r1 = ta[i1 * 4];
            const uint32_t e = te[i4 * 8] ^ te[i3 * 8] ^ te[i2 * 8] ^ te[i1 * 8];
            b0 = b1 ^ r4.x ^ r3.x ^ r2.x ^ r1.x;
            b1 = b2 ^ r4.y ^ r3.y ^ r2.y ^ r1.y;
            b2 = b3 ^ r4.z ^ r3.z ^ r2.z ^ r1.z;
            b3 = b4 ^ r4.w ^ r3.w ^ r2.w ^ r1.w;
            b4 = e;
        }
        const int s = (int)(P % kPackets);
        const uint4 pa = prbs_parity[2 * s];
        const uint32_t pe = reinterpret_cast<const uint32_t*>(prbs_parity)[8 * s + 4];
        parity[2 * P] = make_uint4(b0 ^ pa.x, b1 ^ pa.y, b2 ^ pa.z, b3 ^ pa.w);
        reinterpret_cast<uint32_t*>(parity)[8 * P + 4] = b4 ^ pe;
    }
}

constexpr int kAsmPackets = 64;   // packets per block: 64 * 207 bytes = 3312 dwords, a whole number of dwords
constexpr int kAsmRowDw = 53;

__global__ __launch_bounds__(256) void k_assemble(const uint8_t* __restrict__ ts, size_t ts_stride,
                                                  const uint32_t* __restrict__ parity, uint8_t* __restrict__ coded,
                                                  size_t coded_stride, const uint32_t* __restrict__ prbs188, int fields) {
    __shared__ uint32_t s_pkt[kAsmPackets * kAsmRowDw];  // 13568 B
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c = blockIdx.y;
    const int pkt0 = blockIdx.x * kAsmPackets;
    const int total = fields * kPackets;             // a multiple of 8
    const int npk = min(kAsmPackets, total - pkt0);  // a multiple of 8 as well

    // The block's TS run as flat dwords (coalesced, all loads in flight at once) ^ PRBS, then the 5 parity dwords of
    // every packet, into LDS rows of 53 dwords.  The PRBS repeats every field = 312 * 47 dwords.
    const uint32_t* src = reinterpret_cast<const uint32_t*>(ts + (size_t)c * ts_stride + (size_t)pkt0 * kTsBytes);
    const uint32_t* par = parity + ((size_t)c * total + pkt0) * 8;
    const int ndw = npk * 47;
    uint32_t v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int i = tid + 256 * k;
        const int n = (pkt0 * 47 + i) % (kPackets * 47);
        v[k] = i < ndw ? src[i] ^ prbs188[n] : 0u;
    }
    uint32_t pv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = tid + 256 * k;
        pv[k] = i < npk * 5 ? par[(i / 5) * 8 + i % 5] : 0u;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int i = tid + 256 * k;
        if (i < ndw) s_pkt[i + 6 * (i / 47)] = v[k];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = tid + 256 * k;
        if (i < npk * 5) s_pkt[(i / 5) * kAsmRowDw + 47 + i % 5] = pv[k];
    }
    __syncthreads();

    // Coded packet q = row bytes 1..207.  Four packets = 828 bytes = 207 whole dwords: lane j of a group assembles
    // dword j from bytes 4j..4j+3 of the group, which sit in packet (4j+i)/207 at row byte 1 + (4j+i)%207.
    const uint8_t* s8 = reinterpret_cast<const uint8_t*>(s_pkt);
    uint32_t* dst = reinterpret_cast<uint32_t*>(coded + (size_t)c * coded_stride + kTail + (size_t)pkt0 * kSegCoded);
    int off[4][4];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int beta = 4 * (lane + 64 * it) + i;
            const int p = beta / kSegCoded;
            off[it][i] = p * (kAsmRowDw * 4) + 1 + (beta - p * kSegCoded);
        }
    for (int grp = wv; grp < npk / 4; grp += 4) {
        const uint8_t* g8 = s8 + grp * (4 * kAsmRowDw * 4);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int j = lane + 64 * it;
            if (j < kSegCoded) {
                const uint32_t v = (uint32_t)g8[off[it][0]] | ((uint32_t)g8[off[it][1]] << 8) |
                                   ((uint32_t)g8[off[it][2]] << 16) | ((uint32_t)g8[off[it][3]] << 24);
                dst[grp * kSegCoded + j] = v;
            }
        }
    }
}

void launch_rand_rs(hipStream_t s, const DeviceTables& t, const uint8_t* d_ts, size_t ts_stride, const Buffers& b,
                    int channels, int fields) {
    // ts is [channels][fields][312][188] with ts_stride = fields * 58656: one flat run of packets
    (void)ts_stride;
    const long npackets = (long)channels * fields * kPackets;
    const long want = (npackets + kRsThreads - 1) / kRsThreads;
    const unsigned grid = (unsigned)(want < 1024 ? want : 1024);  // grid-stride: the 96 KB table image loads once per block
    hipLaunchKernelGGL(k_rs_parity, dim3(grid), dim3(kRsThreads), 0, s, d_ts, t.rs_lds, t.prbs_parity, b.parity, npackets);
}

void launch_assemble(hipStream_t s, const DeviceTables& t, const uint8_t* d_ts, size_t ts_stride, const Buffers& b,
                     int channels, int fields) {
    const int total = fields * kPackets;
    hipLaunchKernelGGL(k_assemble, dim3((total + kAsmPackets - 1) / kAsmPackets, channels), dim3(256), 0, s, d_ts, ts_stride,
                       reinterpret_cast<const uint32_t*>(b.parity), b.coded, b.coded_stride, t.prbs188, fields);
}

// trellis helpers (canonical_group, trellis_local): trellis_bits.h
__device__ __forceinline__ uint32_t wave_xor_inclusive(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v ^= o;
    }
    return v;
}

// ======================================================================================= K2 interleaver
// The coded stream of a channel viewed as rows of 52 bytes is sheared: IL[a][r] = E[a - 4r][r]
// (IL[M] = E[M - 208*(M mod 52)], B=52, M=4).  A block produces 621 output rows (half a field) from
// the 42900 coded bytes that start 10608 bytes earlier -- and those are never read from HBM as a coded stream: the
// block stages the ~208 TS packets that hold them (flat coalesced dword loads, PRBS XOR fused: randomize.h:43-54)
// plus their 20 parity bytes from k_rs_parity as LDS rows [sync | 187 data | 20 parity] of 53 dwords, i.e. coded
// byte r of packet q sits at row byte 1 + r.  The four bytes of an output dword lie exactly 207 coded bytes = one
// packet apart, so one (packet, offset) pair per dword addresses all four.  Packets before the batch (first block of
// a batch only) come from the channel's saved 10608-byte coded tail.
// Every thread assembles whole 12-byte groups, so the precoder/trellis parities of the 13 parts of the half-field
// (reference: trellis.h:240-242) are folded in here (LDS XOR) and no separate pass re-reads the interleaved bytes.
constexpr int kIlRows = 621;
constexpr int kIlBytes = kIlRows * 52;               // 32292 output bytes = 2691 groups = 13 parts
constexpr int kIlSrcBytes = (kIlRows + 204) * 52;    // 42900 coded bytes feed them
constexpr int kIlGroups = kIlBytes / 12;             // 2691
constexpr int kIlMaxPk = 210;                        // packets that can overlap 42900 coded bytes
constexpr int kIlRowDw = 53;
constexpr int kIlChPerBlock = 4;                     // channels a block walks with its PRBS slice and gather addresses in registers

typedef uint32_t u32x3a4 __attribute__((ext_vector_type(3), aligned(4)));

__global__ __launch_bounds__(256, 3) void k_interleave(const uint8_t* __restrict__ ts, size_t ts_stride,
                                                    const uint32_t* __restrict__ parity, const uint8_t* __restrict__ tail,
                                                    const uint32_t* __restrict__ prbs188, uint8_t* __restrict__ il,
                                                    size_t il_stride, uint4* __restrict__ par, int channels, int fields,
                                                    int ch_per_block) {
    __shared__ uint32_t s_pkt[kIlMaxPk * kIlRowDw];  // 44520 B
    __shared__ uint32_t s_par[13 * 3][32];           // one XOR accumulator per (part, word, lane mod 32): no same-address traffic
    const int tid = threadIdx.x;
    const int h = blockIdx.x;                          // half-field index within the batch
    const int c_lo = blockIdx.y * ch_per_block, c_hi = min(channels, c_lo + ch_per_block);
    const int total = fields * kPackets;
    const int o0 = h * kIlBytes - kTail;               // coded offset (batch-relative) of the first byte needed
    const int q0 = (o0 + 52 * kSegCoded) / kSegCoded - 52;  // floor(o0 / 207), o0 >= -10608 > -52*207
    const int q1 = (o0 + kIlSrcBytes - 1) / kSegCoded;      // last packet needed (>= 0, < total)
    const int qa = q0 < 0 ? 0 : q0;                    // first packet that lies in this batch
    const int ndw = (q1 - qa + 1) * 47;
    const int npar = (q1 - qa + 1) * 5;
    const int row0 = qa - q0;
    constexpr int kLd4 = (kIlMaxPk * 47 + 1023) / 1024;  // 10 x 4 TS dwords per thread
    constexpr int kLp = (kIlMaxPk * 5 + 255) / 256;    // 5 parity dwords per thread
    constexpr int kIt = (kIlGroups + 255) / 256;       // 11 groups per thread

    // The PRBS slice of the staged packets does not depend on the channel (it repeats every field = 312*47 dwords):
    // read once per block, kept in registers.
    uint32_t prbs[kLd4][4];
#pragma unroll
    for (int k = 0; k < kLd4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * (tid + 256 * k) + j;
            prbs[k][j] = i < ndw ? prbs188[(qa * 47 + i) % (kPackets * 47)] : 0u;
        }
    const int g_field0 = (h & 1) * kIlGroups;          // group index of this block's first group within its field
    const uint8_t* s8 = reinterpret_cast<const uint8_t*>(s_pkt);

    for (int c = c_lo; c < c_hi; ++c) {
        // index math is redone per channel on purpose (opaque copy of tid): hoisted out of the loop it costs ~100 registers
        int tid_c = tid;
        asm volatile("" : "+v"(tid_c));
        for (int i = tid; i < 13 * 3 * 32; i += 256) (&s_par[0][0])[i] = 0;
        // packets qa..q1 of this batch: TS dwords ^ PRBS, then parity; every load is issued before the first LDS store
        {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(ts + (size_t)c * ts_stride) + (size_t)qa * 47;
            const uint32_t* pp = parity + ((size_t)c * total + qa) * 8;
            u32x4a4 v[kLd4];
            uint32_t pv[kLp];
            const int ndw4 = ndw & ~3;                   // ndw is a multiple of 47; the last 0..3 dwords go one by one
#pragma unroll
            for (int k = 0; k < kLd4; ++k) {
                const int i = 4 * (tid_c + 256 * k);
                if (i < ndw4) v[k] = *reinterpret_cast<const u32x4a4*>(src + i);
                else {
                    v[k].x = i < ndw ? src[i] : 0u;
                    v[k].y = i + 1 < ndw ? src[i + 1] : 0u;
                    v[k].z = i + 2 < ndw ? src[i + 2] : 0u;
                    v[k].w = 0u;
                }
            }
#pragma unroll
            for (int k = 0; k < kLp; ++k) {
                const int i = tid_c + 256 * k;
                pv[k] = i < npar ? pp[(i / 5) * 8 + i % 5] : 0u;
            }
#pragma unroll
            for (int k = 0; k < kLd4; ++k) {
                const uint32_t vv[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * (tid_c + 256 * k) + j;
                    if (i < ndw) s_pkt[row0 * kIlRowDw + i + 6 * (i / 47)] = vv[j] ^ prbs[k][j];
                }
            }
#pragma unroll
            for (int k = 0; k < kLp; ++k) {
                const int i = tid_c + 256 * k;
                if (i < npar) s_pkt[(row0 + i / 5) * kIlRowDw + 47 + i % 5] = pv[k];
            }
        }
        // packets before the batch: coded byte o (-10608 <= o < 0) = tail[10608 + o]
        if (q0 < 0) {
            const uint8_t* tl = tail + (size_t)c * kTail;
            uint8_t* s8w = reinterpret_cast<uint8_t*>(s_pkt);
            for (int i = tid; i < kTail; i += 256) {
                const int q = (i + (52 * kSegCoded - kTail)) / kSegCoded - 52;  // floor((i - 10608) / 207)
                const int r = i - kTail - q * kSegCoded;
                s8w[(q - q0) * (kIlRowDw * 4) + 1 + r] = tl[i];
            }
        }
        __syncthreads();

        uint8_t* dst = il + (size_t)c * il_stride + (size_t)h * kIlBytes;
#pragma unroll 1
        for (int it = 0; it < kIt; ++it) {
            const int g = tid_c + 256 * it;
            if (g < kIlGroups) {
                uint32_t dw[3];
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const int d = 3 * g + w;
                    const int a = d / 13;
                    const int r = 4 * (d - 13 * a);
                    const int abs0 = o0 + (a + 204 - 4 * r) * 52 + r;        // coded offset of byte 0; byte j sits 207*j earlier
                    const int q = (abs0 + 52 * kSegCoded) / kSegCoded - 52;   // floor
                    const int addr = (q - q0) * (kIlRowDw * 4) + 1 + (abs0 - q * kSegCoded);
                    dw[w] = (uint32_t)s8[addr] | ((uint32_t)s8[addr - kIlRowDw * 4] << 8) |
                            ((uint32_t)s8[addr - 2 * kIlRowDw * 4] << 16) | ((uint32_t)s8[addr - 3 * kIlRowDw * 4] << 24);
                }
                u32x3a4 o3;
                o3.x = dw[0]; o3.y = dw[1]; o3.z = dw[2];
                // non-temporal: the block's neighbours re-read a third of its TS packets, which the output stream would
                // otherwise push out of the L2 (0.224 -> 0.210 ms per launch)
                __builtin_nontemporal_store(o3, reinterpret_cast<u32x3a4*>(dst + 12 * g));
                // parities of this byte-time for the 12 encoders
                const Words3 e = canonical_group(dw, g_field0 + g);
                const int p = g / kPartGroups;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    uint32_t z2, z0, pr;
                    trellis_local(e.w[k], z2, z0, pr);
                    atomicXor(&s_par[3 * p + k][tid & 31], pr);
                }
            }
        }
        __syncthreads();
        if (tid < 13 * 3) {  // fold the 32 slots (rotated start: conflict-free), then one uint4 per part
            uint32_t v = 0;
#pragma unroll
            for (int i = 0; i < 32; ++i) v ^= s_par[tid][(i + tid) & 31];
            s_par[tid][0] = v;
        }
        __syncthreads();
        if (tid < 13) {
            const int f = h >> 1;
            par[((size_t)c * fields + f) * kParts + (h & 1) * 13 + tid] =
                make_uint4(s_par[3 * tid][0], s_par[3 * tid + 1][0], s_par[3 * tid + 2][0], 0);
        }
        __syncthreads();  // s_pkt and s_par are rewritten for the next channel
    }
}

// Coded tail of the batch (the last 10608 RS-coded bytes) for the next batch's first block and for state export.
__global__ __launch_bounds__(256) void k_save_tail(const uint8_t* __restrict__ ts, size_t ts_stride,
                                                   const uint8_t* __restrict__ parity, const uint8_t* __restrict__ prbs,
                                                   uint8_t* __restrict__ tail, int fields) {
    const int c = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kTail) return;
    const int total = fields * kPackets;
    const int o = total * kSegCoded - kTail + i;   // >= 0: a batch is at least one field
    const int q = o / kSegCoded, r = o - q * kSegCoded;
    const uint8_t v = r < 187 ? (uint8_t)(ts[(size_t)c * ts_stride + (size_t)q * kTsBytes + 1 + r] ^ prbs[(q % kPackets) * 187 + r])
                              : parity[((size_t)c * total + q) * 32 + (r - 187)];
    tail[(size_t)c * kTail + i] = v;
}

void launch_interleave(hipStream_t s, const DeviceTables& t, const uint8_t* d_ts, size_t ts_stride, const Buffers& b,
                       int channels, int fields) {
    const int cpb = kIlChPerBlock;
    dim3 grid(2 * fields, (channels + cpb - 1) / cpb);
    hipLaunchKernelGGL(k_interleave, grid, dim3(256), 0, s, d_ts, ts_stride, reinterpret_cast<const uint32_t*>(b.parity),
                       b.tail, t.prbs188, b.il, b.il_stride, b.par, channels, fields, cpb);
    // the tail is consumed above and replaced by this batch's own (stream order)
    hipLaunchKernelGGL(k_save_tail, dim3((kTail + 255) / 256, channels), dim3(256), 0, s, d_ts, ts_stride,
                       reinterpret_cast<const uint8_t*>(b.parity), t.prbs, b.tail, fields);
}

// ======================================================================================= K3 scan
// One wave per channel: exclusive XOR scan over fields*26 parts, seeded with the channel's carry.
// Lane l owns one field (26 parts) per pass of 64 fields.
__global__ __launch_bounds__(64) void k_trellis_scan(const uint4* __restrict__ par, uint4* __restrict__ cin,
                                                     uint4* __restrict__ carry, uint4* __restrict__ span_par,
                                                     const uint4* __restrict__ carry_xor, int fields) {
    const int lane = threadIdx.x;
    const int c = blockIdx.x;
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

void launch_trellis_scan(hipStream_t s, const Buffers& b, const uint4* carry_xor, int channels, int fields) {
    hipLaunchKernelGGL(k_trellis_scan, dim3(channels), dim3(64), 0, s, b.par, b.cin, b.carry, b.span_par, carry_xor,
                       fields);
}

// XOR of all part parities of the batch, per channel: the span's contribution to the trellis state.
__global__ __launch_bounds__(64) void k_span_parity(const uint4* __restrict__ par, uint4* __restrict__ out, int nparts) {
    const int lane = threadIdx.x;
    const int c = blockIdx.x;
    uint32_t t[3] = {0, 0, 0};
    for (int i = lane; i < nparts; i += 64) {
        const uint4 v = par[(size_t)c * nparts + i];
        t[0] ^= v.x; t[1] ^= v.y; t[2] ^= v.z;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) t[k] ^= __shfl_xor(t[k], d, 64);
    if (lane == 0) out[c] = make_uint4(t[0], t[1], t[2], 0);
}

void launch_span_parity(hipStream_t s, const Buffers& b, uint4* d_out, int channels, int fields) {
    hipLaunchKernelGGL(k_span_parity, dim3(channels), dim3(64), 0, s, b.par, d_out, fields * kParts);
}

// ======================================================================================= K4 map + sync
// Block = one part (12 data segments, 207 byte-times) of one field of one channel.  Lane = byte-time:
// local prefix inside the byte, wave/block XOR scan of the byte parities for the carries, then the
// 4 symbol-times x 12 encoders = 48 symbols of the lane go to an LDS image of the 12 output segments
// (data symbol j of segment s = encoder (4s+j) mod 12, symbol t = 69s + j/12), segment syncs are
// added, and the image is stored with coalesced 16-byte writes.  Part 0 also writes the field-sync
// segment, whose last 12 symbols repeat the previous field's last 12 data symbols: those are
// 4*c2 + 2*x1_last + sB of each encoder at the field boundary, all known from the carries and the
// previous field's last interleaved group.
__global__ __launch_bounds__(256) void k_trellis_map(const uint8_t* __restrict__ il, size_t il_stride,
                                                     const uint4* __restrict__ cin, const uint8_t* __restrict__ last_group,
                                                     const uint8_t* __restrict__ fsync, uint8_t* __restrict__ sym,
                                                     size_t sym_stride, int fields, unsigned parity0) {
    __shared__ uint32_t s_img[12 * kSegSyms / 4];  // 9984 B
    __shared__ uint32_t s_wtot[4][3];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c = blockIdx.z, f = blockIdx.y, p = blockIdx.x;
    const uint8_t* il_f = il + (size_t)c * il_stride + (size_t)f * kFieldCoded;
    const uint4 ci = cin[((size_t)c * fields + f) * kParts + p];
    const uint32_t cinw[3] = {ci.x, ci.y, ci.z};

    uint32_t ew[3] = {0, 0, 0}, z2[3], z0[3], pr[3] = {0, 0, 0};
    const int g = p * kPartGroups + tid;
    if (tid < kPartGroups) {
        const Words3 e = canonical_group(reinterpret_cast<const uint32_t*>(il_f) + 3 * g, g);
#pragma unroll
        for (int k = 0; k < 3; ++k) ew[k] = e.w[k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) trellis_local(ew[k], z2[k], z0[k], pr[k]);

    uint32_t inc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        inc[k] = wave_xor_inclusive(pr[k], lane);
        if (lane == 63) s_wtot[wv][k] = inc[k];
    }
    if (tid < 12) {  // segment syncs 6,1,1,6 of the 12 segments
        s_img[tid * (kSegSyms / 4)] = 0x06010106u;
    }
    __syncthreads();

    if (tid < kPartGroups) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            uint32_t cw = cinw[k] ^ inc[k] ^ pr[k];
            for (int w = 0; w < wv; ++w) cw ^= s_wtot[w][k];
            const uint32_t zz2 = z2[k] ^ ((cw & 0x01010101u) * 0xAAu);
            const uint32_t zz0 = z0[k] ^ (((cw >> 1) & 0x01010101u) * 0x44u) ^ (((cw >> 2) & 0x01010101u) * 0x11u);
            const uint32_t hi = zz2 | (ew[k] & 0x55555555u);
#pragma unroll
            for (int tau = 0; tau < 4; ++tau) {
                const int sh = 6 - 2 * tau;
                const uint32_t v = (((hi >> sh) & 0x03030303u) << 1) | ((zz0 >> sh) & 0x01010101u);
                const int t = 4 * tid + tau;     // symbol time inside the part (0..827)
                const int sl = t / 69;           // local segment 0..11 (field segment 12p+sl, and 12p = 0 mod 3)
                const int q = t - 69 * sl;
                const int hslot = (k + 3 - sl % 3) % 3;
                s_img[sl * (kSegSyms / 4) + 1 + 3 * q + hslot] = v;
            }
        }
    }
    __syncthreads();

    uint8_t* out_f = sym + (size_t)c * sym_stride + (size_t)f * kFieldSyms;
    {
        const uint4* img4 = reinterpret_cast<const uint4*>(s_img);
        uint4* dst = reinterpret_cast<uint4*>(out_f + (size_t)(1 + 12 * p) * kSegSyms);
        for (int i = tid; i < 12 * kSegSyms / 16; i += 256) dst[i] = img4[i];
    }
    if (p == 0 && tid < kSegSyms / 16) {
        // field sync segment; parity of the absolute field index selects the middle PN63 polarity
        const uint4* fs = reinterpret_cast<const uint4*>(fsync + (((parity0 + (unsigned)f) & 1u) ? kSegSyms : 0));
        uint4 v = fs[tid];
        if (tid == 51) {
            // bytes 816..831: 816..819 come from the table, 820..831 are the reserved symbols
            const uint32_t* lg = (f == 0) ? reinterpret_cast<const uint32_t*>(last_group + (size_t)c * 16)
                                          : reinterpret_cast<const uint32_t*>(il_f - 12);
            const int gl = kFieldCoded / 12 - 1;  // last byte-time of the previous field
            const Words3 pe = canonical_group(lg, gl);
            uint32_t rs[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t w = ((cinw[k] & 0x01010101u) << 2) | ((pe.w[k] & 0x01010101u) << 1) |
                                   ((cinw[k] >> 2) & 0x01010101u);
                rs[(k + 1) % 3] = w;  // segment 311: slot h holds encoder group (h + 311) mod 3 = (h + 2) mod 3
            }
            v.y = rs[0]; v.z = rs[1]; v.w = rs[2];
        }
        reinterpret_cast<uint4*>(out_f)[tid] = v;
    }
}

void launch_trellis_map(hipStream_t s, const DeviceTables& t, const Buffers& b, uint8_t* d_sym, size_t sym_stride,
                        int channels, int fields, uint64_t field_base) {
    dim3 grid(kParts, fields, channels);
    hipLaunchKernelGGL(k_trellis_map, grid, dim3(256), 0, s, b.il, b.il_stride, b.cin, b.last_group, t.fsync, d_sym,
                       sym_stride, fields, (unsigned)(field_base & 1));
}

// ======================================================================================= K7 state save
__global__ __launch_bounds__(64) void k_save_state(const uint8_t* __restrict__ il, size_t il_stride,
                                                   uint8_t* __restrict__ last_group, int channels, int fields) {
    const int i = blockIdx.x * 64 + threadIdx.x;  // (channel, dword) of the last 12 interleaved bytes
    const int c = i / 3, w = i - 3 * c;
    if (c >= channels) return;
    const uint32_t* lg = reinterpret_cast<const uint32_t*>(il + (size_t)c * il_stride + (size_t)fields * kFieldCoded - 12);
    reinterpret_cast<uint32_t*>(last_group + (size_t)c * 16)[w] = lg[w];
}

void launch_save_state(hipStream_t s, const Buffers& b, int channels, int fields) {
    hipLaunchKernelGGL(k_save_state, dim3((3 * channels + 63) / 64), dim3(64), 0, s, b.il, b.il_stride, b.last_group,
                       channels, fields);
}

// ======================================================================================= K5 mixer + FIR
// lives in fir_fft.hip (overlap-save fast convolution)

__device__ __forceinline__ float2 osc_step(float2 t, float2 sc) {
    const float t2 = __fmul_rn(t.y, sc.y);
    const float t3 = __fmul_rn(t.x, sc.y);
    return make_float2(__fmaf_rn(t.x, sc.x, -t2), __fmaf_rn(t.y, sc.x, t3));
}

// ======================================================================================= K6 oscillator advance
__global__ __launch_bounds__(256) void k_osc_advance(const float2* __restrict__ in, float2* __restrict__ out,
                                                     float2 scale, unsigned steps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kPaddedSyms) return;
    float2 t = in[i];
    for (unsigned n = 0; n < steps; ++n) t = osc_step(t, scale);
    out[i] = t;
}

void launch_osc_advance(hipStream_t s, const float2* osc_in, float2* osc_out, float2 scale, unsigned steps) {
    hipLaunchKernelGGL(k_osc_advance, dim3((kPaddedSyms + 255) / 256), dim3(256), 0, s, osc_in, osc_out, scale, steps);
}

}  // namespace atsc_k
