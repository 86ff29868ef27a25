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
#include "front_roles.h"
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
// packet's PRBS row is added at the end.  The row lookups are the cost: lanes read random rows; the tables sit in LDS in a
// layout in which no two lanes of a service group can meet (the kernel's own comment), 128 KB, and nothing else lives in
// LDS: 16 waves per CU hide the lookup latency.
//
// k_assemble (debug tap, ATSC_HIP_FLAG_TAPS): coded[207 s + k] = ts[188 s + 1 + k] ^ prbs[187 s + k] (k < 187) followed
// by the 20 parity bytes: a byte-shifting copy, staged through LDS rows of 212 bytes with coalesced loads and stores.
// The path itself never materialises the coded stream: k_interleave reads TS + parity directly.
constexpr int kRsThreads = 1024;

typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));

// PPL: packets per lane, worked on side by side.  The product runs PPL = 1; two independent lookup chains per lane (512
// threads, 169 VGPRs) were measured and lose: 0.103 against 0.091 ms.
template <int PPL>
__global__ __launch_bounds__(kRsThreads / PPL) void k_rs_parity(const uint8_t* __restrict__ ts, const uint4* __restrict__ rs_lds,
                                                                const uint4* __restrict__ prbs_parity,
                                                                uint4* __restrict__ parity, long npackets) {
    constexpr int kThreads = kRsThreads / PPL;
    // The tables in LDS, laid out so that NO lookup conflicts (round 3).  A dword of TS takes four lookups, one per table
    // T1..T4, each a 16-byte part and a 4-byte part of a 20-byte row; the rows are random, so a ds_read_b128 (16 lanes per
    // LDS cycle) is conflict-free only if the 16 lanes of a service group can never meet in a 16-byte slot, and a
    // ds_read_b32 only if the lanes of a half-wave can never meet in a bank.  A copy of a table per lane of a service group
    // would do it (16 x 16 KB: too much); instead the ORDER of a lane's four lookups is rotated by lane: in lookup slot i,
    // lane group g = (lane >> 2) & 3 reads table (g + i) & 3, so the 16 lanes of a service group read four DIFFERENT tables,
    // four lanes each, and four copies per table (copy = lane & 3) give every lane its own slot: row r of table m, copy c at
    // byte r * 256 + 64 m + 16 c -- 64 KB (the service groups of ds_read_b128 are lanes {0-3, 12-15, 20-27}, {4-11, 16-19,
    // 28-31}, ... : each holds every (lane group, copy) pair once).  The 4-byte parts the same way: a ds_read_b32 serves lanes
    // 0-31 and 32-63 in one LDS cycle each over 32 banks, so a half-wave needs 32 different banks: dword 8 m + 2 c + (lane bit
    // 4) of the row's first 32 dwords for lanes 0-31, the same + 32 for lanes 32-63 -- another 64 KB.  XOR is commutative, so
    // the order of the lookups is free.
    // An address is ONE v_perm_b32: byte (3 - m) of u into bits 8..15 over a per-lane base.
    __shared__ uint4 s_tab[8192];            // [A: 256 rows x 16 slots x 16 B | E: 256 rows x 64 banks x 4 B] = 128 KB
    const int tid = threadIdx.x;
    {
        const uint32_t* e_src = reinterpret_cast<const uint32_t*>(rs_lds) + 16384;   // host image: dword (m*256 + v)*8 (tables.cpp)
#pragma unroll
        for (int it = 0; it < 4 * PPL; ++it) {
            const int j = tid + kThreads * it;
            const int r = j >> 4, m = (j >> 2) & 3, me = (j >> 1) & 3;   // 16-byte slot j & 15: table of the A part / of the E dwords
            s_tab[j] = rs_lds[(m * 256 + r) * 4];
            const uint32_t e = e_src[(me * 256 + r) * 8];
            s_tab[4096 + j] = make_uint4(e, e, e, e);
        }
    }
    __syncthreads();

    const int lane = tid & 63;
    const uint32_t c4 = lane & 3, g = (lane >> 2) & 3, k4 = (lane >> 4) & 3;
    uint32_t sel[4], base_a[4], base_e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t m = (g + i) & 3;
        // table index m = 3 is T4 = the first byte in time = the lowest byte of the dword: table m takes byte 3 - m
        sel[i] = 0x0c020000u | ((4u + (3u - m)) << 8);          // {0, base byte 2, u byte (3 - m), base byte 0}
        base_a[i] = 64u * m + 16u * c4;
        base_e[i] = 0x10000u + 4u * (8u * m + 2u * c4 + (k4 & 1u) + 32u * (k4 >> 1));
    }
    const uint8_t* lds = reinterpret_cast<const uint8_t*>(s_tab);
    // A block takes kRsThreads consecutive packets per trip (lane's packets are kThreads apart).  (Loading the next trip's
    // dwords into the registers of this trip as they fall free -- no bubble at the top of a trip, no extra registers -- was
    // measured and loses: 0.107 against 0.091 ms; the loads then sit between the lookups of every fourth step.)
    for (long P0 = (long)blockIdx.x * kRsThreads + tid; P0 < npackets; P0 += (long)gridDim.x * kRsThreads) {
        uint32_t d[PPL][48];
        bool ok[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const long P = P0 + (long)q * kThreads;
            ok[q] = P < npackets;
            const uint8_t* pk = ts + (ok[q] ? P : P0) * kTsBytes;
            const u32x4a4* src = reinterpret_cast<const u32x4a4*>(pk);
#pragma unroll
            for (int i = 0; i < 11; ++i) {
                const u32x4a4 v = src[i];
                d[q][4 * i] = v.x; d[q][4 * i + 1] = v.y; d[q][4 * i + 2] = v.z; d[q][4 * i + 3] = v.w;
            }
            const uint32_t* tail = reinterpret_cast<const uint32_t*>(pk) + 44;
            d[q][44] = tail[0]; d[q][45] = tail[1]; d[q][46] = tail[2];
            d[q][0] &= 0xffffff00u;  // TS sync byte: dropped unchecked (randomize.h:48)
        }
        uint32_t b0[PPL], b1[PPL], b2[PPL], b3[PPL], b4[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) b0[q] = b1[q] = b2[q] = b3[q] = b4[q] = 0;
#pragma unroll
        for (int w = 0; w < 47; ++w) {
            uint4 r[PPL][4];
            uint32_t e[PPL][4];
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                const uint32_t u = d[q][w] ^ b0[q];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    r[q][i] = *reinterpret_cast<const uint4*>(lds + __builtin_amdgcn_perm(u, base_a[i], sel[i]));
                    e[q][i] = *reinterpret_cast<const uint32_t*>(lds + __builtin_amdgcn_perm(u, base_e[i], sel[i]));
                }
            }
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                b0[q] = b1[q] ^ r[q][0].x ^ r[q][1].x ^ r[q][2].x ^ r[q][3].x;
                b1[q] = b2[q] ^ r[q][0].y ^ r[q][1].y ^ r[q][2].y ^ r[q][3].y;
                b2[q] = b3[q] ^ r[q][0].z ^ r[q][1].z ^ r[q][2].z ^ r[q][3].z;
                b3[q] = b4[q] ^ r[q][0].w ^ r[q][1].w ^ r[q][2].w ^ r[q][3].w;
                b4[q] = e[q][0] ^ e[q][1] ^ e[q][2] ^ e[q][3];
            }
        }
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const long P = P0 + (long)q * kThreads;
            if (!ok[q]) continue;
            const int s = (int)(P % kPackets);
            const uint4 pa = prbs_parity[2 * s];
            const uint32_t pe = reinterpret_cast<const uint32_t*>(prbs_parity)[8 * s + 4];
            // The record is shifted by three bytes (parity byte j at byte 3 + j): the interleaver keeps a packet's 207 coded
            // bytes dword-aligned in LDS, where the first parity byte is the top byte of the dword that holds data bytes 184..186
            const uint32_t p0 = b0[q] ^ pa.x, p1 = b1[q] ^ pa.y, p2 = b2[q] ^ pa.z, p3 = b3[q] ^ pa.w, p4 = b4[q] ^ pe;
            parity[2 * P] = make_uint4(p0 << 24, __builtin_amdgcn_alignbyte(p1, p0, 1), __builtin_amdgcn_alignbyte(p2, p1, 1),
                                       __builtin_amdgcn_alignbyte(p3, p2, 1));
            reinterpret_cast<uint2*>(parity)[4 * P + 2] = make_uint2(__builtin_amdgcn_alignbyte(p4, p3, 1), p4 >> 8);
        }
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
        if (i < npk * 5) {   // dword (i % 5) of the packet's 20 parity bytes: record bytes 3 + 4 (i % 5) ...
            const uint32_t* rec = par + (i / 5) * 8 + i % 5;
            pv[k] = __builtin_amdgcn_alignbyte(rec[1], rec[0], 3);
        } else {
            pv[k] = 0u;
        }
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
    // One workgroup per CU, grid-stride: the 128 KB table image is built once per CU (1024 workgroups, four rounds of table
    // builds: 0.102 ms; 512: 0.096; 256: 0.091, same box)
    static int cus_of[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && cus_of[dev] == 0) {
        int n = 0;
        cus_of[dev] = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0 ? n : 256;
    }
    const long cap = dev >= 0 && dev < 64 ? cus_of[dev] : 256;
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    hipLaunchKernelGGL(k_rs_parity<1>, dim3(grid), dim3(kRsThreads), 0, s, d_ts, t.rs_lds, t.prbs_parity, b.parity, npackets);
}

void launch_assemble(hipStream_t s, const DeviceTables& t, const uint8_t* d_ts, size_t ts_stride, const Buffers& b,
                     int channels, int fields) {
    const int total = fields * kPackets;
    hipLaunchKernelGGL(k_assemble, dim3((total + kAsmPackets - 1) / kAsmPackets, channels), dim3(256), 0, s, d_ts, ts_stride,
                       reinterpret_cast<const uint32_t*>(b.parity), b.coded, b.coded_stride, t.prbs188, fields);
}

// trellis helpers (canonical_group, trellis_local): trellis_bits.h; wave_xor_inclusive: front_roles.h

// ======================================================================================= K2 interleaver
// IL[M] = E[M - 208 (M mod 52)]  (B = 52, M = 4), E the channel's continuous RS-coded stream.
// 52 bytes per output row and 207 bytes per coded packet meet every 10764 bytes = 207 output rows = 52 packets: one CHUNK.
// Writing an output position of chunk k as (row a < 207, r < 52) and the coded stream in packets,
//
//     IL[k][a][r] = packet (52 k + qa - r), byte ta,      with 52 a = 207 qa + ta
//
// i.e. an output ROW is a COLUMN of the packet matrix, read upwards over 52 consecutive packets; a -> ta is a bijection of
// 0..206 (52 * 4 = 1 mod 207), and only the packets of chunks k - 1 and k are touched.  So a block keeps the packets of two
// chunks in LDS (rows of 53 dwords, a packet's 207 coded bytes dword-aligned: TS bytes 1..187 come in shifted by one byte,
// XORed with the PRBS, the 20 parity bytes arrive pre-shifted from k_rs_parity) and walks the chunks of whole fields of one
// channel in order: every TS byte is read once (+ one chunk per block of fields), and the packets of chunk k stay for k + 1.
// The gather works on dwords: lane = (dword column c, group of four rows rho) reads the SAME dword column of seven
// consecutive packet rows and forms four output dwords with byte permutes (the four columns of a dword belong to four
// output rows whose row windows are one packet apart) -- 7 LDS dword reads per 16 output bytes instead of 16 byte reads.
// The output dwords go to an LDS image of the chunk and leave as 16-byte stores; the pass that stores them also folds in
// the precoder / trellis parities of the 26 parts of a field (reference: trellis.h:240-242), per dword: a dword of the
// interleaved stream holds the bytes of four encoders of one byte-time.
constexpr int kCkPk = 52;                          // packets per chunk
constexpr int kCkRows = 207;                       // output rows per chunk
constexpr int kCkBytes = kCkPk * kSegCoded;        // 10764
constexpr int kCkDw = kCkBytes / 4;                // 2691
constexpr int kCkTs = kCkPk * kTsBytes;            // 9776 TS bytes per chunk
constexpr int kCkPerField = kPackets / kCkPk;      // 6
constexpr int kIlRowDw = 53;                       // LDS row of a packet: 52 dwords + 1 (odd stride)
constexpr int kIlRing = 2 * kCkPk;                 // 104 packet rows: two chunks, halves alternate
constexpr int kIlMirror = 6;                       // rows 104..109 repeat rows 0..5: a window of 7 rows never wraps
constexpr int kIlTasks = kCkPk * 13;               // 676 = (packet, 16-byte piece) when staging = (dword column, 4 rows) when gathering
constexpr int kIlPieces = (kCkDw + 3) / 4;         // 673 sixteen-byte pieces leave per chunk
constexpr int kIlSegs = kCkBytes / kSegCoded;      // 52 segments of 207 interleaved bytes per chunk (12 to a part)
// Eight waves: a workgroup then puts two waves on every SIMD whatever SIMD its first wave lands on, so up to three workgroups
// (the register budget: 80 VGPRs = 6 waves per SIMD) always fit a CU.  With six waves the fourth workgroup often did not fit
// (one SIMD already held six waves) and the grid ran in two rounds.
constexpr int kIlThreads = 512;
constexpr int kIlIts = (kIlTasks + kIlThreads - 1) / kIlThreads;
static_assert(kIlIts == 2, "two tasks per thread");
constexpr int kIlPkPerWave = 5;                                              // TS staging: five packets x 11 pieces = 55 lanes of a wave
constexpr int kIlTsWaves1 = (kCkPk - 8 * kIlPkPerWave + kIlPkPerWave - 1) / kIlPkPerWave;   // waves with a second TS task: 3
constexpr int kIlBWave = kIlTsWaves1, kIlCWave = kIlTsWaves1 + 1;           // waves whose second task is a packet's tail / parity
constexpr int kIlGatherWaves1 = (kIlTasks - kIlThreads + 63) / 64;          // waves with a second gather task
constexpr int kIlStoreWaves1 = (kIlPieces - kIlThreads + 63) / 64;          // waves with a second piece to store
constexpr int kIlParTid0 = 64 * (kIlCWave + 1);                              // the parity threads: the three waves behind those
constexpr int kIlFoldTid0 = 64 * kIlCWave;                                   // the wave with the lightest second task folds
static_assert(kIlThreads == 512 && kIlParTid0 + kIlSegs * 3 <= kIlThreads, "wave roles");

typedef uint32_t u32x4a1 __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32x2a1 __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t u32x1a1 __attribute__((aligned(1)));
typedef uint32_t u32x4a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef uint32_t u32x3a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int V> struct IntC { static constexpr int value = V; };
#define ATSC_GLOBAL __attribute__((address_space(1)))
// a wave-uniform address as a scalar register pair: scalar base + 32-bit lane offset is the addressing mode wanted
__device__ __forceinline__ uint64_t scalar_addr(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

__global__ __launch_bounds__(kIlThreads) void k_interleave(const uint8_t* __restrict__ ts, size_t ts_stride,
                                                           const uint8_t* __restrict__ parity, const uint8_t* __restrict__ tail,
                                                           uint8_t* __restrict__ tail_next,
                                                           const uint8_t* __restrict__ prbs_rows, uint8_t* __restrict__ il,
                                                           size_t il_stride, uint32_t* __restrict__ par, int fields,
                                                           int fields_per_block) {
    __shared__ uint32_t s_pkt_mem[1 + (kIlRing + kIlMirror) * kIlRowDw];   // 23324 B: a guard dword, then the rows
    uint32_t* const s_pkt = s_pkt_mem + 1;
    __shared__ __align__(16) uint32_t s_out[kCkDw + 5];            // the chunk in transmission order
    __shared__ uint32_t s_par[16 * 3];                             // XOR of the data dwords per (part mod 16, word)
    const int tid = threadIdx.x;
    const int c = blockIdx.y;
    const int f0 = blockIdx.x * fields_per_block, f1 = min(fields, f0 + fields_per_block);
    const uint8_t* ts_c = ts + (size_t)c * ts_stride;
    const uint8_t* rec_c = parity + (size_t)c * fields * (kPackets * 32);
    uint8_t* il_c = il + (size_t)c * il_stride;
    if (tid < 16 * 3) s_par[tid] = 0;

    // This kernel is bound by instruction issue, not by a memory: everything that does not change from chunk to chunk
    // (task geometry, LDS addresses, offsets from the chunk's base pointers) is computed here, once.
    // ---- gather tasks: id = 13 hi + lo -> dword column hi (coded bytes 4 hi .. 4 hi + 3 = output rows a0 + 4 j), output
    // bytes 4 lo .. 4 lo + 3 of those rows
    bool g_ok[kIlIts], g_all4[kIlIts];
    const uint32_t* g_src[kIlIts][2];
    uint32_t* g_dst[kIlIts];
#pragma unroll
    for (int it = 0; it < kIlIts; ++it) {
        const int id = tid + kIlThreads * it;
        g_ok[it] = id < kIlTasks;
        const int hi = id / 13, lo = id - 13 * hi;
        g_all4[it] = hi < 51;        // column 207 does not exist
        const int a0 = (16 * hi) % kCkRows;
        const int first = kCkPk + (52 * a0) / kSegCoded - 4 * lo - 3;   // logical row (0..103: previous chunk, this chunk)
        // chunk kk lives in half kk & 1: for even kk the previous chunk is rows 52..103 and logical row l is (l + 52) mod 104
        g_src[it][0] = s_pkt + (first >= kCkPk ? first - kCkPk : first + kCkPk) * kIlRowDw + hi;
        g_src[it][1] = s_pkt + first * kIlRowDw + hi;
        g_dst[it] = s_out + a0 * 13 + lo;
    }
    // ---- staging tasks: 16 bytes of TS or parity each.  A vector memory instruction costs its pipe the same with 4 lanes
    // active as with 64, and a load that is not dword-aligned costs it double (measured: 10 % of this kernel), so the loads are
    // aligned and the one-byte shift between TS and coded bytes (coded byte t of a packet is TS byte t + 1, randomize.h:48) is
    // done in registers; the kinds of task are sorted into whole waves:
    //   TS task (packet q, piece p < 11), 55 lanes = five packets per wave, both passes: TS dwords 4p .. 4p+3 -> coded dwords
    //       4p-1 .. 4p+2 (the byte in front comes from the lane below; for p = 0 the dword in front of the row is the padding
    //       dword of the row before it, or the guard dword)
    //   second task of wave kIlBWave, packet q = lane: TS dwords 43..46 -> coded dwords 43..46 with the first parity byte, and
    //       parity bytes 1..4 (parity record dwords 0, 1)
    //   second task of wave kIlCWave, packet q = lane: parity bytes 5..19 (record dwords 2..5)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // The lanes of a wave issue the same loads (no lane-dependent branches between them: a merge of registers after such a
    // branch makes the compiler wait for the data there, which would serialise the prefetch); lanes without a task load
    // offset 0 and drop it.
    bool st_ok[2], st_mirror[2];
    uint32_t a_off[2], pr_off[2], rec_off = 0;
    uint32_t* st_dst[2][2];
    {
        int q[2], p[2];
        const int lq = lane / 11, lp = lane - 11 * lq;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            q[it] = kIlPkPerWave * (wave + 8 * it) + lq;
            p[it] = lp;
            st_ok[it] = lq < kIlPkPerWave && q[it] < kCkPk && (it == 0 || wave < kIlTsWaves1);
            if (!st_ok[it]) q[it] = p[it] = 0;
            a_off[it] = q[it] * kTsBytes + 16 * p[it];
            pr_off[it] = q[it] * 208 + 16 * p[it];          // relative to four bytes in front of the chunk's PRBS rows (fetch)
            st_dst[it][0] = s_pkt + q[it] * kIlRowDw + 4 * p[it] - 1;
        }
        if (wave == kIlBWave || wave == kIlCWave) {
            st_ok[1] = lane < kCkPk;
            q[1] = st_ok[1] ? lane : 0;
            // B: TS bytes 172..187 + parity record dwords 0, 1; C: record dwords 2..5
            a_off[1] = wave == kIlBWave ? q[1] * kTsBytes + 172 : q[1] * 32 + 8;
            pr_off[1] = q[1] * 208 + 4 + (wave == kIlBWave ? 172 : 192);    // row bytes 187..207 of the table are zero
            rec_off = q[1] * 32;
            st_dst[1][0] = s_pkt + q[1] * kIlRowDw + (wave == kIlBWave ? 43 : 48);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            st_mirror[it] = q[it] < kIlMirror;
            st_dst[it][1] = st_dst[it][0] + kCkPk * kIlRowDw;
        }
    }
    // ---- leaving: 16-byte piece id of the chunk
    const uint32_t out_off0 = 16 * tid, out_off1 = 16 * (tid + kIlThreads);
    const uint4* out_src0 = reinterpret_cast<const uint4*>(s_out) + tid;
    const uint4* out_src1 = tid + kIlThreads < kIlPieces ? out_src0 + kIlThreads : out_src0;
    // ---- trellis parities: thread (segment sg, dword w of a 12-byte group) of the chunk's 52 x 3
    const int pt = tid - kIlParTid0;
    const bool pt_ok = pt >= 0 && pt < kIlSegs * 3;
    const int sg = pt_ok ? pt / 3 : 0, sw = pt_ok ? pt - 3 * sg : 0;
    const int sg_first = (kSegCoded * sg + 11) / 12;
    const bool sg_18 = (kSegCoded * (sg + 1) + 11) / 12 - sg_first == 18;   // 17 or 18 groups
    const uint32_t* sg_src = s_out + 3 * sg_first + sw;

    u32x4a1 va[2];
    u32x4 vb[2];
    u32x2 vr;
    auto fetch = [&](int kk) {
        const uint64_t tsk = scalar_addr(reinterpret_cast<uint64_t>(ts_c) + (uint64_t)kk * kCkTs);
        const uint64_t reck = scalar_addr(reinterpret_cast<uint64_t>(rec_c) + (uint64_t)kk * (kCkPk * 32));
        // Coded dwords 4p-1.. of a TS task start four bytes in front of the PRBS piece 16p: the base is moved, not the unsigned
        // lane offsets.  For the first row of the table those four bytes are the end of the table in front of it in the same
        // allocation (atsc_hip.cpp: off_prr >= 256); the value lands in a padding dword.
        const uint64_t prk = scalar_addr(reinterpret_cast<uint64_t>(prbs_rows) + (uint32_t)(kk % kCkPerField) * (kCkPk * 208) - 4);
        const uint64_t base1 = wave == kIlCWave ? reck : tsk;
        va[0] = *reinterpret_cast<const ATSC_GLOBAL u32x4a1*>(reinterpret_cast<const ATSC_GLOBAL uint8_t*>(tsk) + a_off[0]);
        vb[0] = *reinterpret_cast<const ATSC_GLOBAL u32x4a4*>(reinterpret_cast<const ATSC_GLOBAL uint8_t*>(prk) + pr_off[0]);
        if (wave <= kIlCWave) {      // (wave-uniform)
            va[1] = *reinterpret_cast<const ATSC_GLOBAL u32x4a1*>(reinterpret_cast<const ATSC_GLOBAL uint8_t*>(base1) + a_off[1]);
            vb[1] = *reinterpret_cast<const ATSC_GLOBAL u32x4a4*>(reinterpret_cast<const ATSC_GLOBAL uint8_t*>(prk) + pr_off[1]);
            vr = *reinterpret_cast<const ATSC_GLOBAL u32x2*>(reinterpret_cast<const ATSC_GLOBAL uint8_t*>(reck) + rec_off);
        }
    };
    // TS task: loaded dwords d0..d3 = TS dwords 4p..4p+3 -> coded dwords 4p-1..4p+2, each the upper three bytes of one TS dword and
    // the lowest byte of the next; `below` is d3 of the lane below (the piece in front)
    auto ts_shift = [&](const u32x4a1& v, uint32_t below, uint32_t (&x)[4]) {
        x[0] = __builtin_amdgcn_alignbyte(v.x, below, 1);
        x[1] = __builtin_amdgcn_alignbyte(v.y, v.x, 1);
        x[2] = __builtin_amdgcn_alignbyte(v.z, v.y, 1);
        x[3] = __builtin_amdgcn_alignbyte(v.w, v.z, 1);
    };
    auto commit = [&](auto half_c) {
        constexpr int H = decltype(half_c)::value;
        const uint32_t below0 = __shfl_up(va[0].w, 1, 64), below1 = __shfl_up(va[1].w, 1, 64);   // (every lane takes part)
        {
            uint32_t x[4];
            ts_shift(va[0], below0, x);
            if (st_ok[0]) {
                uint32_t* dst = st_dst[0][H];
                const uint32_t x0 = x[0] ^ vb[0].x, x1 = x[1] ^ vb[0].y, x2 = x[2] ^ vb[0].z, x3 = x[3] ^ vb[0].w;
                dst[0] = x0; dst[1] = x1; dst[2] = x2; dst[3] = x3;
                if (H == 0 && st_mirror[0]) {
                    dst += kIlRing * kIlRowDw;
                    dst[0] = x0; dst[1] = x1; dst[2] = x2; dst[3] = x3;
                }
            }
        }
        if (wave < kIlTsWaves1) {
            uint32_t x[4];
            ts_shift(va[1], below1, x);
            if (st_ok[1]) {
                uint32_t* dst = st_dst[1][H];
                const uint32_t x0 = x[0] ^ vb[1].x, x1 = x[1] ^ vb[1].y, x2 = x[2] ^ vb[1].z, x3 = x[3] ^ vb[1].w;
                dst[0] = x0; dst[1] = x1; dst[2] = x2; dst[3] = x3;
                if (H == 0 && st_mirror[1]) {
                    dst += kIlRing * kIlRowDw;
                    dst[0] = x0; dst[1] = x1; dst[2] = x2; dst[3] = x3;
                }
            }
        } else if (wave == kIlBWave) {
            // TS dwords 43..46 -> coded dwords 43..46 (TS bytes 173..187 and, in the top byte of the last one, the first parity
            // byte, which k_rs_parity left there), then parity bytes 1..4
            if (st_ok[1]) {
                uint32_t* dst = st_dst[1][H];
                const uint32_t x0 = __builtin_amdgcn_alignbyte(va[1].y, va[1].x, 1) ^ vb[1].x,
                               x1 = __builtin_amdgcn_alignbyte(va[1].z, va[1].y, 1) ^ vb[1].y,
                               x2 = __builtin_amdgcn_alignbyte(va[1].w, va[1].z, 1) ^ vb[1].z,
                               x3 = ((va[1].w >> 8) ^ vb[1].w) | vr.x, x4 = vr.y;
                dst[0] = x0; dst[1] = x1; dst[2] = x2; dst[3] = x3; dst[4] = x4;
                if (H == 0 && st_mirror[1]) {
                    dst += kIlRing * kIlRowDw;
                    dst[0] = x0; dst[1] = x1; dst[2] = x2; dst[3] = x3; dst[4] = x4;
                }
            }
        } else if (wave == kIlCWave) {
            if (st_ok[1]) {
                uint32_t* dst = st_dst[1][H];
                dst[0] = va[1].x; dst[1] = va[1].y; dst[2] = va[1].z; dst[3] = va[1].w;
                if (H == 0 && st_mirror[1]) {
                    dst += kIlRing * kIlRowDw;
                    dst[0] = va[1].x; dst[1] = va[1].y; dst[2] = va[1].z; dst[3] = va[1].w;
                }
            }
        }
    };

    // ---- the chunk before the block's first one (odd: half 1): from the batch, or from the coded tail of the previous batch
    const int kk_first = f0 * kCkPerField, kk_end = f1 * kCkPerField;
    if (kk_first > 0) {
        fetch(kk_first - 1);
        commit(IntC<1>());
    } else {
        // the last 51 packets before the batch (= the last 10557 bytes of the tail) -> rows 1..51 of half 1, as unaligned
        // dwords with every load in flight at once: 64 of the grid's blocks come through here, and a slow copy here is the
        // kernel's tail (the last dword of a row reads three bytes of the next packet into the row's padding; the very
        // last one would read past the tail and is assembled from bytes)
        uint32_t* rows = s_pkt + (kCkPk + 1) * kIlRowDw;
        const uint8_t* tl = tail + (size_t)c * kTail + (kTail - 51 * kSegCoded);
        constexpr int kTailDw = 51 * 52;
        constexpr int kTailIts = (kTailDw + kIlThreads - 1) / kIlThreads;
        uint32_t tv[kTailIts];
#pragma unroll
        for (int it = 0; it < kTailIts; ++it) {
            const int i = tid + kIlThreads * it;
            const int q = i / 52, w = i - 52 * q;
            tv[it] = 0;
            if (i < kTailDw - 1) tv[it] = *reinterpret_cast<const u32x1a1*>(tl + q * kSegCoded + 4 * w);
            else if (i == kTailDw - 1) tv[it] = tl[50 * kSegCoded + 204] | (tl[50 * kSegCoded + 205] << 8) | (tl[50 * kSegCoded + 206] << 16);
        }
#pragma unroll
        for (int it = 0; it < kTailIts; ++it) {
            const int i = tid + kIlThreads * it;
            const int q = i / 52, w = i - 52 * q;
            if (i < kTailDw) rows[q * kIlRowDw + w] = tv[it];
        }
    }
    fetch(kk_first);
    commit(IntC<0>());
    __syncthreads();

    // one chunk: kk in the batch, kf in its field (kf & 1 == H)
    // The parts that ended inside chunk kf of field f are complete once that chunk's barrier has passed: one wave (the one
    // whose second staging task is the lightest) turns them into parities.  It does so one chunk LATER, beside the other
    // waves' store pass (measured with cycle-counter probes: done right behind the barrier it made this wave 1500 cycles
    // late at the next one, where seven waves waited);
    // the slots are indexed by part mod 16 so that the parts of the following chunk never land on one still to be folded.
    auto fold = [&](int kf, int f) {
        const int done_before = (kf * kIlSegs) / 12, done_after = ((kf + 1) * kIlSegs) / 12;
        const int ft = tid - kIlFoldTid0;
        if (ft >= 0 && ft < (done_after - done_before) * 3) {
            const int pi = ft / 3, k = ft - 3 * pi;
            const int slot = ((done_before + pi) & 15) * 3 + k;
            uint32_t* o = par + (((size_t)c * fields + f) * kParts + done_before + pi) * 4;
            uint32_t z2, z0, pr;
            trellis_local(s_par[slot], z2, z0, pr);
            o[k] = pr;
            if (k == 0) o[3] = 0;
            s_par[slot] = 0;
        }
    };
    auto chunk = [&](int kk, int kf, int f, auto half_c) {
        constexpr int H = decltype(half_c)::value;
        const bool more = kk + 1 < kk_end;
        if (more) fetch(kk + 1);      // in flight during the gather
        // ---- gather: seven rows of one dword column -> four output dwords (byte j of rows first + 3 + j - e, e = 0..3)
#pragma unroll
        for (int it = 0; it < kIlIts; ++it) {
            if (it == 1 && wave >= kIlGatherWaves1) break;      // (wave-uniform)
            if (g_ok[it]) {
                const uint32_t* src = g_src[it][H];
                uint32_t rw[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) rw[i] = src[i * kIlRowDw];
                uint32_t* dst = g_dst[it];
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    constexpr uint32_t kPick = 0x0c0c0400u;   // byte 0 <- second operand, byte 1 <- first operand; + j: byte j of each
                    const uint32_t lo2 = __builtin_amdgcn_perm(rw[2 + j], rw[3 + j], kPick + 0x0101u * j);
                    const uint32_t hi2 = __builtin_amdgcn_perm(rw[j], rw[1 + j], kPick + 0x0101u * j);
                    o[j] = lo2 | (hi2 << 16);
                }
                dst[0] = o[0]; dst[52] = o[1]; dst[104] = o[2];
                if (g_all4[it]) dst[156] = o[3];
            }
        }
        __syncthreads();
        if (kk > kk_first) fold(kf ? kf - 1 : kCkPerField - 1, kf ? f : f - 1);
        // ---- the chunk leaves as 16-byte pieces
        const uint64_t dst = scalar_addr(reinterpret_cast<uint64_t>(il_c) + (uint64_t)kk * kCkBytes);
        {
            const uint4 v = *out_src0;
            u32x4a4 o;
            o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
            *reinterpret_cast<ATSC_GLOBAL u32x4a4*>(reinterpret_cast<ATSC_GLOBAL uint8_t*>(dst) + out_off0) = o;
        }
        if (wave < kIlStoreWaves1) {
            const uint4 v = *out_src1;       // the pieces behind the first kIlThreads (s_out is padded)
            if (out_off1 < 16 * (kIlPieces - 1)) {
                u32x4a4 o;
                o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
                *reinterpret_cast<ATSC_GLOBAL u32x4a4*>(reinterpret_cast<ATSC_GLOBAL uint8_t*>(dst) + out_off1) = o;
            } else if (out_off1 == 16 * (kIlPieces - 1)) {
                u32x3a4 o;
                o.x = v.x; o.y = v.y; o.z = v.z;
                *reinterpret_cast<ATSC_GLOBAL u32x3a4*>(reinterpret_cast<ATSC_GLOBAL uint8_t*>(dst) + out_off1) = o;
            }
        }
        // ---- precoder / trellis parities of the chunk's 52 segments (reference: trellis.h:240-242).  The parity is linear in
        // the data bits, so the raw dwords are XORed here, per (part, 4-encoder word), and the parity is taken once per part:
        // within a segment the encoder assignment does not rotate (trellis_bits.h)
        if (pt_ok) {
            uint32_t acc = sg_18 ? sg_src[3 * 17] : 0u;
#pragma unroll
            for (int i = 0; i < 17; ++i) acc ^= sg_src[3 * i];
            const int r = 4 * kf + sg;                       // segment of the field = 48 kf + r, r < 72
            const int S = kf * kIlSegs + sg;
            atomicXor(&s_par[((4 * kf + r / 12) & 15) * 3 + (sw + S) % 3], acc);
        }
        if (more) commit(IntC<H ^ 1>());
        __syncthreads();
    };
    int kk = kk_first;
#pragma unroll 1
    for (int f = f0; f < f1; ++f) {
#pragma unroll 1
        for (int kf = 0; kf < kCkPerField; kf += 2) {
            chunk(kk, kf, f, IntC<0>());
            chunk(kk + 1, kf + 1, f, IntC<1>());
            kk += 2;
        }
    }
    fold(kCkPerField - 1, f1 - 1);      // (behind the last chunk's closing barrier)
    // The workgroup that ends the channel's batch leaves the coded tail for the next batch (and for state export): the last
    // 10608 RS-coded bytes = the last 51 packets and the last 51 bytes of the one before, all of them rows of the last chunk,
    // which sits in half 1 of the ring (an odd chunk; nothing was committed behind it).  It goes to the OTHER tail buffer --
    // the workgroup of the channel's first fields may still be reading the current one -- and the host swaps the two.
    if (f1 == fields) {
        const uint8_t* rows = reinterpret_cast<const uint8_t*>(s_pkt + kCkPk * kIlRowDw);
        uint32_t* dst = reinterpret_cast<uint32_t*>(tail_next + (size_t)c * kTail);
        constexpr int kFirst = kCkPk * kSegCoded - kTail;      // 156: offset of the tail in the chunk's coded bytes
        for (int j = tid; j < kTail / 4; j += kIlThreads) {
            uint32_t v = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = kFirst + 4 * j + i;
                const int row = o / kSegCoded, b = o - row * kSegCoded;
                v |= (uint32_t)rows[row * (kIlRowDw * 4) + b] << (8 * i);
            }
            dst[j] = v;
        }
    }
}

void launch_interleave(hipStream_t s, const DeviceTables& t, const uint8_t* d_ts, size_t ts_stride, const Buffers& b,
                       int channels, int fields) {
    // Workgroups of whole fields (a part's parity must not be shared between workgroups).  Measured at 64 x 64 fields: two
    // workgroups per CU (512 in all, 8 fields each) 0.153 ms, three 0.164, one 0.22 -- a CU gets through a chunk in about
    // 1.5 us however many workgroups share it, and one workgroup alone needs 2.3 us per chunk.  Never more than
    // 16 fields per workgroup (it re-reads one chunk in front of its fields).
    constexpr int kIlGridTarget = 512;
    int fpb = (int)(((long)channels * fields + kIlGridTarget - 1) / kIlGridTarget);
    fpb = fpb < 1 ? 1 : (fpb > 16 ? 16 : fpb);
    dim3 grid((fields + fpb - 1) / fpb, channels);
    // (the batch's own coded tail is written into b.tail_next by the workgroup that ends each channel; the caller swaps)
    hipLaunchKernelGGL(k_interleave, grid, dim3(kIlThreads), 0, s, d_ts, ts_stride, reinterpret_cast<const uint8_t*>(b.parity),
                       b.tail, b.tail_next, t.prbs_rows, b.il, b.il_stride, reinterpret_cast<uint32_t*>(b.par), fields, fpb);
}

// ======================================================================================= K3 scan
// One wave per channel (front_roles.h: trellis_scan_body).
__global__ __launch_bounds__(64) void k_trellis_scan(const uint4* __restrict__ par, uint4* __restrict__ cin,
                                                     uint4* __restrict__ carry, uint4* __restrict__ span_par,
                                                     const uint4* __restrict__ carry_xor, int fields) {
    trellis_scan_body(par, cin, carry, span_par, carry_xor, fields, blockIdx.x, threadIdx.x);
}

// The fused path's three small launches in one grid (front_roles.h): workgroups [0, channels) scan a channel's parities
// (first wave), the next ones fill the oscillator table of the batch, the last ones copy the batch's last interleaved bytes
// into the NEXT history buffer (the FIR behind this launch still reads the current one).
__global__ __launch_bounds__(256) void k_scan_osc_state(const uint4* __restrict__ par, uint4* __restrict__ cin,
                                                        uint4* __restrict__ carry, uint4* __restrict__ span_par,
                                                        const uint4* __restrict__ carry_xor, int channels, int fields,
                                                        const float2* __restrict__ osc_in, float2* __restrict__ osc_tab,
                                                        float2* __restrict__ osc_out, float2 sc, int planes,
                                                        const uint8_t* __restrict__ il, size_t il_stride,
                                                        uint8_t* __restrict__ last_group_next) {
    constexpr int kOscBlocks = (kPaddedSyms / 2 + 255) / 256;
    const int b = blockIdx.x;
    if (b < channels) {
        if (threadIdx.x < 64) trellis_scan_body(par, cin, carry, span_par, carry_xor, fields, b, threadIdx.x);
    } else if (b < channels + kOscBlocks) {
        osc_table_body(osc_in, osc_tab, osc_out, sc, fields, planes, (b - channels) * 256 + threadIdx.x);
    } else {
        save_last_group_body(il, il_stride, last_group_next, channels, fields, (b - channels - kOscBlocks) * 256 + threadIdx.x);
    }
}

void launch_scan_osc_state(hipStream_t s, const Buffers& b, const uint4* carry_xor, int channels, int fields, const float2* osc_in,
                           float2* osc_out, float2* osc_tab, float2 scale, int planes) {
    constexpr int kOscBlocks = (kPaddedSyms / 2 + 255) / 256;
    const unsigned grid = (unsigned)(channels + kOscBlocks + (3 * channels + 255) / 256);
    hipLaunchKernelGGL(k_scan_osc_state, dim3(grid), dim3(256), 0, s, b.par, b.cin, b.carry, b.span_par, carry_xor, channels, fields,
                       osc_in, osc_tab, osc_out, scale, planes, b.il, b.il_stride, b.last_group_next);
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
                                                   uint8_t* __restrict__ last_group_next, int channels, int fields) {
    save_last_group_body(il, il_stride, last_group_next, channels, fields, blockIdx.x * 64 + threadIdx.x);
}

void launch_save_state(hipStream_t s, const Buffers& b, int channels, int fields) {
    hipLaunchKernelGGL(k_save_state, dim3((3 * channels + 63) / 64), dim3(64), 0, s, b.il, b.il_stride, b.last_group_next,
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
