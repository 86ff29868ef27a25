// gfx950 pilot/offset mixer + RRC FIR on the matrix cores, with and without the trellis / map / sync stage fused in.
//
//   y[c][i] = sum_{k<111} taps[k] * ( T_f[i+k] * x_c[i+k] )       (reference: offset.h:11-45, filter.h:9-69)
//
// The mixer runs FIRST, as in the reference: z_c[m] = x_c[m] * T_f[m] is one single-rounded fp32 product per component
// (x is real: offset.h:24-31).  The filter is then a product with a CONSTANT banded matrix:
//
//   Y[p][n] = sum_k A[p][k] * Z[k][n],     A[p][k] = taps[k - p]  (Toeplitz, 32 outputs x 144 inputs),
//                                          Z[k][n] = z of channel n/2, component n%2, at input i0 + k
//
// i.e. a GEMM with M = 32 consecutive outputs, N = 32 columns = 16 channels x (re, im), K = 144 = nine K-steps of
// v_mfma_f32_32x32x16_f16.  Both operands are split into two fp16 pieces (hi + lo, scaled into fp16's normal range):
// the taps once on the host -- the A operand sits in registers for the whole kernel, nothing is regenerated per tile --
// and z per element: level (exact in fp16) times the oscillator, which k_osc_table hands over already split into two fp16
// pieces, in three packed fp16 instructions per PAIR of elements (mix_pair below).  Y = Ahi Zhi + Alo Zhi + Ahi Zlo with fp32
// accumulation carries >= 21 significant bits of every product (the dropped Alo Zlo term is 2^-22 relative).
// A tile advances by 32 outputs = two K-steps, so the B operand (Z) is a sliding window of nine K-step slots in
// registers: a tile converts only its 32 newest inputs per column, the rest is a renaming (the tile loop is unrolled
// by nine).  What the previous form of this kernel spent per 16 outputs x 32 channels on regenerating its weights
// (160 vector instructions: the oscillator was folded into the A operand) is gone; per 32 outputs x 16 channels a wave
// now issues 21 MFMAs (27 products less the low-order ones of the outer K-steps) and ~50 vector instructions for the operands.
//
// One wave = one (field, run of tiles, group of 16 channels) task:
//   * symbols reach the wave through a wave-private LDS ring [16 channels][512 positions], stored as v << 5 so that a
//     byte becomes an fp16 level with one v_perm and one v_pk_fma.  FUSED: the ring is filled by the wave's own
//     precoder / trellis / mapper / sync producer from the interleaved bytes (reference: trellis.h:10-51,
//     field_sync.h:12-26; 0.25 B per symbol from HBM, the symbols never exist in HBM).  Not FUSED: coalesced 16-byte
//     loads of the u8 symbols that k_trellis_map wrote, three tiles ahead.
//   * the oscillator reaches it as one dword per lane and tile (32 positions x re/im), a whole round of nine tiles at a
//     time, through LDS windows double-buffered by round parity; the batch table T_f comes from k_osc_table (re / im
//     planes of split entries).
//   * C (32x32 fp32) goes through a wave-private LDS image [channel][32 outputs] in the OUTPUT sample format and leaves
//     as whole 128-byte lines, 16 bytes per lane, spread over the next tile's MFMA sequence.
// No workgroup barriers; waves are independent (LDS operations of a wave execute in program order).
#include <hip/hip_ext.h>

#include "iq_format.h"
#include "kernels.h"
#include "trellis_bits.h"

#include <vector>

namespace atsc_k {

namespace {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

constexpr int kTile = 32;                           // outputs per tile = two MFMA K-steps
constexpr int kTiles = kFieldSyms / kTile;          // 8138 tiles per field
constexpr int kKSteps = 9;                          // 144 >= 32 + 110 inputs feed a tile
constexpr int kRound = 9;                           // tiles per unrolled round = period of the slot renaming
constexpr int kAhead = 3;                           // tiles between a prefetch and its use (divides kRound)
constexpr int kWaves = 4;
constexpr int kRing = 512;                          // positions per channel row of the symbol ring
constexpr int kRingRow = kRing + 16;                // bytes: 132 dwords = 4 mod 64 -> the 16 rows' b64 reads are conflict-free
constexpr int kByteTimes = kFieldCoded / 12;           // 5382 byte-times per field, 207 per part
constexpr float kAScale = 65536.0f;                 // taps are scaled into fp16's normal range ...
constexpr float kZScale = 4096.0f;                  // ... and so are the mixed samples (|z| <= 7.5); 2^-28 undoes both, exactly
constexpr int kSymShift = 5;                        // ring bytes are v << 5: fp16 (0x6400 | byte) = 1024 + 32 v

static_assert(kFieldSyms % kTile == 0, "tiles must cover a field exactly");
static_assert(kTile + 110 <= 16 * kKSteps, "K must cover the filter");
static_assert(kRound % kAhead == 0 && (2 * kRound) % kKSteps == 0, "static register renaming");
static_assert(kFusedGroup == 16, "one wave task = 16 channels");

// A operand: [phase][hi | lo][K-step][lane] x 16 bytes, in the MFMA lane layout (lane = output row p + 32 * half, 8
// consecutive k).  Phase 0: the reference's taps; phase 1: the same pulse half a symbol later (2x output, see UP below)
__device__ uint4 g_a_tab[2 * 2 * kKSteps * 64];

template <int FMT, int UP> struct Geo {
    typedef typename OutSample<FMT>::type out_t;
    static constexpr int kRowBytes = kTile * UP * (int)sizeof(out_t);   // one channel's samples of a tile: 256 / 128 / 64 B x UP
    static constexpr int kImgRow = kRowBytes + 16;                   // LDS image row (16-byte aligned, rows on different banks)
    static constexpr int kPieces = kRowBytes / 16;                   // 16-byte pieces per row
    static constexpr int kRowsPerStore = 64 / kPieces;               // channel rows one store instruction covers: 4 / 8 / 16
    static constexpr int kStores = kFusedGroup / kRowsPerStore;      // store instructions per tile: 4 / 2 / 1
};

__device__ __forceinline__ void lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Two ring bytes -> two fp16 levels 2v - 5.75 (signal.h:13-18): 0x6400 | (v << 5) is 1024 + 32 v, times 1/16 minus 69.75.
__device__ __forceinline__ uint32_t levels2(uint32_t ring_dword, bool upper) {
    const h2 k16 = {(_Float16)0.0625f, (_Float16)0.0625f}, kb = {(_Float16)-69.75f, (_Float16)-69.75f};
    const uint32_t pr = __builtin_amdgcn_perm(0x64646464u, ring_dword, upper ? 0x04030402u : 0x04010400u);
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, pr) * k16 + kb);
}

// The mixer product z = level * T (offset.h:24-31) as two fp16 pieces, for two inputs at once, in three full-rate packed
// fp16 instructions.  The oscillator arrives already split, T * 4096 = Th + Tl (two fp16 pieces per table entry, made once
// per position by k_osc_table: the split does not depend on the channel):
//   hi = fl16(lv * Th)                      v_pk_mul_f16
//   r  = lv * Th - hi                       v_pk_fma_f16: the rounding error of an fp16 product is an fp16 number, so r is exact
//   lo = fl16(lv * Tl + r)                  v_pk_fma_f16
// hi + lo = level * T to 2^-21 relative (the level has 6 significant bits, Th + Tl carries 22 of T's 24).  The first version
// formed the fp32 product and split it with v_fma_mixlo/hi_f16: 4 instructions per pair that each cost 2.6 x a plain one.
__device__ __forceinline__ void mix_pair(uint32_t lv, uint32_t th, uint32_t tl, uint32_t& hi, uint32_t& lo) {
    const h2 l = __builtin_bit_cast(h2, lv), a = __builtin_bit_cast(h2, th), b = __builtin_bit_cast(h2, tl);
    const h2 p = l * a;
    const h2 r = __builtin_elementwise_fma(l, a, -p);
    hi = __builtin_bit_cast(uint32_t, p);
    lo = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(l, b, r));
}

// Two pairs (one ring dword), the two dependency chains interleaved: a packed fp16 instruction that reads the result of the
// one just before it costs a wait state.
__device__ __forceinline__ void mix_quad(uint32_t dw, const uint4& t, uint32_t& h0, uint32_t& l0, uint32_t& h1, uint32_t& l1) {
#if defined(__HIP_DEVICE_COMPILE__)   // (written out: hipcc's scheduler re-orders the C form back into two serial chains)
    const uint32_t k64 = 0x64646464u, sel01 = 0x04010400u, sel23 = 0x04030402u;
    const uint32_t k16 = 0x2c002c00u /* 1/16, 1/16 */, kb = 0xd45cd45cu /* -69.75, -69.75 */;
    uint32_t la, lb;
    asm("v_perm_b32 %4, %6, %7, %8\n\t"
        "v_perm_b32 %5, %6, %7, %9\n\t"
        "v_pk_fma_f16 %4, %4, %10, %11\n\t"
        "v_pk_fma_f16 %5, %5, %10, %11\n\t"
        "v_pk_mul_f16 %0, %4, %12\n\t"
        "v_pk_mul_f16 %2, %5, %14\n\t"
        "v_pk_fma_f16 %1, %4, %12, %0 neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
        "v_pk_fma_f16 %3, %5, %14, %2 neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
        "v_pk_fma_f16 %1, %4, %13, %1\n\t"
        "v_pk_fma_f16 %3, %5, %15, %3"
        : "=&v"(h0), "=&v"(l0), "=&v"(h1), "=&v"(l1), "=&v"(la), "=&v"(lb)
        : "v"(k64), "v"(dw), "v"(sel01), "v"(sel23), "v"(k16), "v"(kb), "v"(t.x), "v"(t.y), "v"(t.z), "v"(t.w));
#else
    h0 = l0 = h1 = l1 = 0;
#endif
}

// One K-step of the B operand for this lane: 8 consecutive inputs of its (channel, component) column.
// sy: 8 ring bytes; t0, t1: the table entries of the 4 position pairs {Th pair, Tl pair} x 2 each
__device__ __forceinline__ void make_kstep(uint2 sy, uint4 t0, uint4 t1, uint4& bhi, uint4& blo) {
    mix_pair(levels2(sy.x, false), t0.x, t0.y, bhi.x, blo.x);
    mix_pair(levels2(sy.x, true), t0.z, t0.w, bhi.y, blo.y);
    mix_pair(levels2(sy.y, false), t1.x, t1.y, bhi.z, blo.z);
    mix_pair(levels2(sy.y, true), t1.z, t1.w, bhi.w, blo.w);
}

// one component of a sample in the output format (the pair form lives in iq_format.h)
template <int FMT>
__device__ __forceinline__ void put_component(uint8_t* img_row, int pos, int comp, float y) {
    if constexpr (FMT == kIqCf32) {
        reinterpret_cast<float*>(img_row)[2 * pos + comp] = y;
    } else if constexpr (FMT == kIqCs16) {
        reinterpret_cast<uint16_t*>(img_row)[2 * pos + comp] = (uint16_t)to_out<kIqCs16>((v2f_t){y, 0.0f});
    } else {
        img_row[2 * pos + comp] = (uint8_t)to_out<kIqCs8>((v2f_t){y, 0.0f});
    }
}

struct FirArgs {
    // symbols: the source when not fused (u8 level indices from k_trellis_map), the optional tap when fused
    uint8_t* sym;
    size_t sym_stride;
    void* iq;
    size_t iq_stride;          // samples
    const float* planes;       // [field][re plane | im plane] of kPaddedSyms floats (k_osc_table, planes form)
    int channels, fields, rounds, runs_per_field, ch_groups;
    // fused only
    const uint8_t* il;
    size_t il_stride;
    const uint4* cin;
    const uint8_t* last_group;
    const uint8_t* fsync;
    unsigned parity0;
};

}  // namespace

// UP = 2: the upsampling form of the filter ("upsampling to the SDR rate").  The same pulse evaluated at half-symbol
// offsets is a second Toeplitz operand A1 on the SAME Z window (112 taps centred between taps 55 and 56 fit the 144 inputs):
//   y2[2 i] = sum_k taps[k] z[i + k]   (the reference's sample, bit for bit the UP = 1 output)
//   y2[2 i + 1] = sum_k taps_half[k] z[i + k]
// i.e. a polyphase interpolator by 2 whose phase 0 is the reference filter; one pass over the symbols, no 1x IQ in HBM.  54
// MFMAs per tile on two accumulators, the Z operand is converted once; both A operands stay in registers, so one wave per
// SIMD (and the kernel becomes matrix-pipe- and store-bound: twice the products, twice the bytes).
// Which of the three fp16 products a K-step needs.  Y = Ahi.Zhi + Alo.Zhi + Ahi.Zlo; the two low-order products correct the
// rounding of the taps (|Alo| <= 2^-11 |tap|, and the remainders are constants) and of z (|Zlo| <= 2 * 2^-11 |z|: the
// oscillator's split and the product's rounding).  A K-step holds taps k - p for its 16 inputs k and the tile's 32 outputs p,
// and the taps of the outer K-steps are small: the largest sum over a row of K-steps 0, 1, 7, 8 is 0.0006, 0.0016, 0.0014,
// 0.0005 of sum |tap| = 0.189.  The tap correction is left out of those four and the z correction out of K-steps 0 and 8,
// where the product takes z rounded ONCE (fl16(Zhi + Zlo), four packed adds per K-step) in place of Zhi.  What that and
// every other term costs, whatever the input (|z| <= 8.25 * 0.9, worst output row, every rounding at its bound and aligned;
// scripts/fir_error_budget.py computes it from this operand's own bits, tests/test_cabi_cpu.py asserts it):
//     tap correction left out 2.4e-6 + z correction left out 2.3e-6 + z split residual 1.0e-6 + Alo.Zlo 1.5e-7
//     + accumulation along the 21 MFMAs 1.3e-6 (charged as a chain of correctly rounded fp32 operations)
//         = 7.2e-6 of full scale   (half-symbol phase: 8.6e-6)
// inside the 1e-5 the path is held to; seen: 1.7e-6 on 1e9 random samples, 3.0e-6 on symbols chosen against exactly these
// omissions (tests/test_gpu_round3.py: the error at the targets is the one the float64 restatement aims for, to 3e-7).  It
// saves 6 of the tile's 27 MFMAs -- the kernel runs at the board's power cap, where time follows energy.
// Skipping the z correction in K-steps 1 and 7 as well (19 MFMAs, another 2 % faster) has a worst case of 1.3e-5 and was
// not taken.
constexpr bool tap_lo_needed(int ks) { return ks >= 2 && ks <= 6; }
constexpr bool z_lo_needed(int ks) { return ks >= 1 && ks <= 7; }
constexpr int products_of(int ks) { return 1 + (tap_lo_needed(ks) ? 1 : 0) + (z_lo_needed(ks) ? 1 : 0); }
constexpr int steps_in(int ks, int first, int last) { return ks <= first ? 0 : (ks > last ? last - first + 1 : ks - first); }
constexpr int chain_before(int ks) {   // MFMAs of one output phase in front of K-step ks
    return ks + steps_in(ks, 2, 6) + steps_in(ks, 1, 7);
}
static_assert(chain_before(kKSteps) == 21 && chain_before(2) == 3, "21 MFMAs per tile and phase");

template <int FMT, bool FUSED, int UP>
__global__ __launch_bounds__(64 * kWaves, UP == 1 ? 2 : 1) void k_mix_fir_mm(const FirArgs a) {
    typedef Geo<FMT, UP> G;
    typedef typename G::out_t out_t;
    __shared__ __align__(16) uint8_t s_ring[kWaves][kFusedGroup * kRingRow];
    __shared__ __align__(16) uint8_t s_img[kWaves][kFusedGroup * G::kImgRow];
    __shared__ __align__(16) float s_osc[kWaves][2 * kRound * 64];   // [round parity][tile of the round]: [re | im] x 32 newest inputs

    // the wave index is made scalar: task, field, tile and every loop bound below are then wave-uniform SGPR values
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, h = lane >> 5;
    const int cl = n >> 1, comp = n & 1;                  // this lane's column of Z: channel cl of the group, re / im
    const long task = (long)blockIdx.x * kWaves + wv;
    const int run = (int)(task % a.runs_per_field);
    const long gidx = task / a.runs_per_field;
    const int cg = (int)(gidx % a.ch_groups);
    const int f = (int)(gidx / a.ch_groups);
    if (f >= a.fields) return;
    const int tile0 = run * a.rounds * kRound;
    const int P0 = tile0 * kTile;
    const int rows_valid = min(kFusedGroup, a.channels - cg * kFusedGroup);   // channel rows of this group that exist
    const int ch = cg * kFusedGroup + min(cl, rows_valid - 1);                // rows past the end repeat the last one; never stored

    // ---- A operand: the Toeplitz tap matrix, constant, in registers (read from LDS per K-step it cost 0.2 ms per launch)
    h8 ahi[UP][kKSteps], alo[UP][kKSteps];
#pragma unroll
    for (int ph = 0; ph < UP; ++ph)
#pragma unroll
        for (int ks = 0; ks < kKSteps; ++ks) {
            ahi[ph][ks] = __builtin_bit_cast(h8, g_a_tab[((2 * ph) * kKSteps + ks) * 64 + lane]);
            alo[ph][ks] = __builtin_bit_cast(h8, g_a_tab[((2 * ph + 1) * kKSteps + ks) * 64 + lane]);
        }

    // ---- memory: buffer descriptors with a wave-uniform base, a per-lane 32-bit offset and a scalar per-tile offset, so
    // that no vector instruction is spent on addresses; stores to rows or tiles that do not exist are dropped by the
    // range check (rows past the end lie behind num_records; for a tile that does not exist num_records is made 0)
    constexpr int kRsrcFlags = 0x00020000;   // raw buffer, 32-bit data format
    const uint32_t iq_row_bytes = (uint32_t)(a.iq_stride * sizeof(out_t));
    uint8_t* iq_base = reinterpret_cast<uint8_t*>(a.iq) + ((size_t)cg * kFusedGroup * a.iq_stride + (size_t)f * kFieldSyms * UP) * sizeof(out_t);
    const int iq_records = (int)((uint32_t)rows_valid * iq_row_bytes);
    const int st_piece = lane % G::kPieces, st_row = lane / G::kPieces;
    const uint32_t st_off = (uint32_t)st_row * iq_row_bytes + 16u * st_piece;               // + q * kRowsPerStore rows
    const uint32_t st_step = (uint32_t)G::kRowsPerStore * iq_row_bytes;
    const __amdgpu_buffer_rsrc_t rs_osc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.planes) + (size_t)f * 2 * kPaddedSyms, 0, 2 * kPaddedSyms * 4, kRsrcFlags);
    const int osc_off = ((lane >> 5) * kPaddedSyms + (lane & 31)) * 4;   // lanes 0..31 the re plane, 32..63 the im plane
    auto osc_fetch = [&](int tile) {   // the 32 newest inputs of `tile`; tiles past the field stay inside the padded table
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_osc, osc_off, ((tile < kTiles ? tile : 0) * kTile + 112) * 4, 0));
    };

    uint8_t* ring = s_ring[wv];
    const uint8_t* ring_rd = ring + cl * kRingRow + 8 * h;                    // + (position & 511)
    uint8_t* img = s_img[wv];
    uint8_t* img_wr = img + cl * G::kImgRow;                                  // this lane's channel row
    const uint8_t* img_rd = img + st_row * G::kImgRow + 16 * st_piece;        // + q * kRowsPerStore rows
    float* oscw = s_osc[wv];
    const float* osc_rd = oscw + comp * 32 + 8 * h;                           // + 64 u, + 16 s, + j
    float* osc_wr = oscw + lane;                                              // + 64 u: lane brings one float of a tile's window

    // ---- symbol source when not fused: lanes 0..31 bring 16 bytes of a channel row each (row = lane / 2)
    __amdgpu_buffer_rsrc_t rs_sym = rs_osc;
    int sym_off = 0;
    uint8_t* ring_wr = nullptr;
    if constexpr (!FUSED) {
        const int row = (lane >> 1) & 15;
        rs_sym = __builtin_amdgcn_make_buffer_rsrc(a.sym + (size_t)cg * kFusedGroup * a.sym_stride + (size_t)f * kFieldSyms, 0,
                                                   (int)((uint32_t)rows_valid * (uint32_t)a.sym_stride), kRsrcFlags);
        sym_off = (int)((uint32_t)min(row, rows_valid - 1) * (uint32_t)a.sym_stride);
        ring_wr = ring + row * kRingRow;
    }
    auto sym_fetch = [&](int tile) {   // 16 of the 32 newest inputs of `tile` (sync-head copy past the field end, field_sync.h:16)
        const int pos = (tile < kTiles ? tile : 0) * kTile + 112 + 16 * (lane & 1);
        const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs_sym, sym_off + (pos < kFieldSyms ? pos : pos - kFieldSyms), 0, 0);
        return make_uint4(v.x << kSymShift, v.y << kSymShift, v.z << kSymShift, v.w << kSymShift);
    };
    auto sym_park = [&](int tile, const uint4& v) {
        *reinterpret_cast<uint4*>(ring_wr + ((tile * kTile + 112 + 16 * (lane & 1)) & (kRing - 1))) = v;
    };

    // ---- FUSED: the symbol producer.  Lane = (channel pc of the group, byte-time slot pq): one step turns four consecutive
    // byte-times (12 interleaved bytes = 48 data symbols each) of the 16 channels into ring symbols -- precoder, trellis
    // encoder, mapper, segment syncs exactly as k_trellis_map (kernels.hip; reference trellis.h:10-51, field_sync.h:12-26);
    // the four lanes of a channel chain their byte-times through two shuffles of the byte parities.
    const int pq = lane >> 4, pc = lane & 15;
    const int pch = cg * kFusedGroup + min(pc, rows_valid - 1);
    const uint8_t* il_c = FUSED ? a.il + (size_t)pch * a.il_stride + (size_t)f * kFieldCoded : nullptr;
    uint8_t* pring = ring + pc * kRingRow;
    int next_g = 0;        // next byte-time of the field to produce (uniform)
    int fs_j = -1;         // next field-sync symbol to produce, or -1 when not inside the field-sync segment
    int fs_base = 0;       // 0: the field's own sync segment; kFieldSyms: its copy behind the field (field_sync.h:16)
    int prod_pos = 0;      // every (extended) position below it is in the ring
    int tap_pos = 0;       // next 128-symbol block of the symbol tap
    uint32_t cw[3] = {0, 0, 0}, rsv[3] = {0, 0, 0};
    const uint32_t* fs_tab = nullptr;
    __amdgpu_buffer_rsrc_t rs_tap = rs_osc;
    if constexpr (FUSED) {
        fs_tab = reinterpret_cast<const uint32_t*>(a.fsync + (((a.parity0 + (unsigned)f) & 1u) ? kSegSyms : 0));
        if (a.sym)
            rs_tap = __builtin_amdgcn_make_buffer_rsrc(a.sym + (size_t)cg * kFusedGroup * a.sym_stride + (size_t)f * kFieldSyms, 0,
                                                       (int)((uint32_t)rows_valid * (uint32_t)a.sym_stride), kRsrcFlags);
        const uint4* cin_c = a.cin + ((size_t)pch * a.fields + f) * kParts;
        // where the run starts
        int part0;
        const int seg = P0 / kSegSyms, jj = P0 - seg * kSegSyms;
        if (seg == 0) {
            next_g = 0; fs_j = jj; part0 = 0;
        } else {
            const int sg = seg - 1, pp = sg / 12, sl = sg - 12 * pp;
            const int j = jj > 4 ? jj - 4 : 0;
            next_g = kPartGroups * pp + ((69 * sl + j / 12) >> 2);
            part0 = pp;   // the byte-time that holds P0 is produced first; earlier symbols of it fall on dead ring slots
        }
        prod_pos = P0;
        tap_pos = (P0 + 127) & ~127;
        // trellis state entering next_g: the part's carry and the parities of the part's earlier byte-times
        {
            const uint4 ci = cin_c[part0];
            uint32_t px[3] = {0, 0, 0};
            for (int gq = part0 * kPartGroups + pq; gq < next_g; gq += 32) {   // eight byte-times per lane in flight: latency-bound
                uint32_t d[8][3];
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) {
                    const uint32_t* src = reinterpret_cast<const uint32_t*>(il_c) + 3 * min(gq + 4 * k8, kByteTimes - 1);
                    d[k8][0] = src[0]; d[k8][1] = src[1]; d[k8][2] = src[2];
                }
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) {
                    const int gg = gq + 4 * k8;
                    if (gg < next_g) {
                        const Words3 e = canonical_group(d[k8], gg);
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            uint32_t z2, z0, pr;
                            trellis_local(e.w[k], z2, z0, pr);
                            px[k] ^= pr;
                        }
                    }
                }
            }
            const uint32_t civ[3] = {ci.x, ci.y, ci.z};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                px[k] ^= __shfl_xor(px[k], 16, 64);
                px[k] ^= __shfl_xor(px[k], 32, 64);
                cw[k] = civ[k] ^ px[k];
            }
        }
        // reserved symbols of the field-sync segment (field_sync.h:17,23): the last 12 data symbols of the previous field
        {
            const uint4 c0 = cin_c[0];
            const uint32_t cinw[3] = {c0.x, c0.y, c0.z};
            const uint32_t* lg = (f == 0) ? reinterpret_cast<const uint32_t*>(a.last_group + (size_t)pch * 16)
                                          : reinterpret_cast<const uint32_t*>(il_c - 12);
            const Words3 pe = canonical_group(lg, kByteTimes - 1);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                rsv[(k + 1) % 3] = ((cinw[k] & 0x01010101u) << 2) | ((pe.w[k] & 0x01010101u) << 1) | ((cinw[k] >> 2) & 0x01010101u);
        }
    }
    auto il_fetch = [&](int g, uint32_t& d0, uint32_t& d1, uint32_t& d2) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(il_c) + 3 * min(g + pq, kByteTimes - 1);
        d0 = src[0]; d1 = src[1]; d2 = src[2];
    };
    // four byte-times: slot pq takes next_g + pq, whose three interleaved dwords are d0..d2
    auto produce_quad = [&](uint32_t d0, uint32_t d1, uint32_t d2) {
        const int g = next_g + pq;
        const bool ok = g < kByteTimes;
        const int gg = ok ? g : kByteTimes - 1;
        const uint32_t d3[3] = {d0, d1, d2};
        const Words3 e = canonical_group(d3, gg);
        uint32_t hi[3], zz0[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            uint32_t z2, z0, pr;
            trellis_local(e.w[k], z2, z0, pr);
            if (!ok) pr = 0;
            const uint32_t nb = __shfl_xor(pr, 16, 64);          // the pair's other byte-time
            const uint32_t pair = pr ^ nb;
            const uint32_t other = __shfl_xor(pair, 32, 64);      // the other pair
            const uint32_t cwk = cw[k] ^ ((pq & 1) ? nb : 0u) ^ ((pq & 2) ? other : 0u);   // parities of the earlier slots
            cw[k] ^= pair ^ other;
            const uint32_t zz2 = z2 ^ ((cwk & 0x01010101u) * 0xAAu);
            zz0[k] = z0 ^ (((cwk >> 1) & 0x01010101u) * 0x44u) ^ (((cwk >> 2) & 0x01010101u) * 0x11u);
            hi[k] = zz2 | (e.w[k] & 0x55555555u);
        }
        if (ok) {
            // the four symbol times of the byte: segment, place in the segment and commutator phase once per symbol time,
            // then the three 4-encoder words go to slots (k - phase) mod 3; ring bytes are v << 5
            const int pp = gg / kPartGroups, tq = gg - pp * kPartGroups;
            const int t0 = 4 * tq;
            int sl = t0 / 69, q = t0 - 69 * sl;
            int m = sl % 3;
#pragma unroll
            for (int tau = 0; tau < 4; ++tau) {
                const int seg = 1 + 12 * pp + sl;
                const int base = kSegSyms * seg + 4 + 12 * q;
                if (q == 0) *reinterpret_cast<uint32_t*>(pring + ((kSegSyms * seg) & (kRing - 1))) = 0x06010106u << kSymShift;
                const int shb = 6 - 2 * tau;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint32_t v = (((hi[k] >> shb) & 0x03030303u) << (1 + kSymShift)) | (((zz0[k] >> shb) & 0x01010101u) << kSymShift);
                    int hslot = k - m;
                    hslot += hslot < 0 ? 3 : 0;
                    *reinterpret_cast<uint32_t*>(pring + ((base + 4 * hslot) & (kRing - 1))) = v;
                }
                if (++q == 69) { q = 0; ++sl; m = m == 2 ? 0 : m + 1; }
            }
        }
        next_g += 4;
        if (next_g >= kByteTimes) {
            fs_j = 0; fs_base = kFieldSyms; prod_pos = kFieldSyms;    // behind the field: the copy of its sync head
        } else {
            // first data symbol of byte-time next_g, less the segment sync that may precede it
            const int pp = next_g / kPartGroups, t = 4 * (next_g - pp * kPartGroups), sl = t / 69;
            prod_pos = kSegSyms * (1 + 12 * pp + sl) + 4 + 12 * (t - 69 * sl) - 4;
        }
    };
    // symbol tap: every complete 128-symbol block below prod_pos -> d_sym, whole 128-byte lines, two stores per block
    const int tap_end = P0 + kRound * kTile * a.rounds - 1;   // the run taps the blocks that start inside it
    auto tap_blocks = [&](int upto) {
        if (!a.sym) return;
        const int end = min(min(prod_pos, upto + 127), kFieldSyms + 127) & ~127;
        while (tap_pos < end) {
            const int pp = tap_pos + 16 * (lane & 7);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int row = 8 * q + (lane >> 3);
                const uint8_t* src = ring + row * kRingRow + (pp & (kRing - 1));
                const uint2 x = *reinterpret_cast<const uint2*>(src), y = *reinterpret_cast<const uint2*>(src + 8);
                const uint32_t off = pp < kFieldSyms ? (uint32_t)row * (uint32_t)a.sym_stride + (uint32_t)pp : 0xffffffffu;
                __builtin_amdgcn_raw_buffer_store_b128((v4u){x.x >> kSymShift, x.y >> kSymShift, y.x >> kSymShift, y.y >> kSymShift},
                                                       rs_tap, (int)off, 0, 2);
            }
            tap_pos += 128;
        }
    };
    // everything below `limit` (extended position) into the ring
    auto produce_until = [&](int limit) {
        while (prod_pos < limit) {
            if (fs_j >= 0) {
                // field-sync segment: the table, its last three dwords replaced by the reserved symbols; the four lanes of a
                // channel write every fourth dword
                const int nsym = min(min(kSegSyms - fs_j, ((limit - prod_pos) + 3) & ~3), 256);   // (the tap keeps up in steps)
                for (int d = (fs_j >> 2) + pq; d < ((fs_j + nsym) >> 2); d += 4) {
                    const uint32_t v = d < 205 ? fs_tab[d] : (d == 205 ? rsv[0] : (d == 206 ? rsv[1] : rsv[2]));
                    *reinterpret_cast<uint32_t*>(pring + ((fs_base + 4 * d) & (kRing - 1))) = v << kSymShift;
                }
                fs_j += nsym;
                prod_pos += nsym;
                if (fs_j >= kSegSyms) {
                    fs_j = -1;
                    if (fs_base) prod_pos = 1 << 30;     // the copy behind the field is complete: nothing further exists
                    else prod_pos = kSegSyms;             // data segment 1 follows (its sync comes with byte-time 0)
                }
                tap_blocks(tap_end);
                continue;
            }
            uint32_t d0, d1, d2;
            il_fetch(next_g, d0, d1, d2);
            produce_quad(d0, d1, d2);
            tap_blocks(tap_end);
        }
    };
    if constexpr (FUSED) {
        produce_until(P0 + kKSteps * 16);
        lds_order();
    }

    // ---- prime: all nine K-steps of the run's first tile = inputs [P0, P0 + 144)
    uint4 bhi[kKSteps], blo[kKSteps];
    {
        const float* tp = a.planes + ((size_t)f * 2 + comp) * kPaddedSyms + P0 + 8 * h;
        const uint8_t* sp = a.sym + (size_t)ch * a.sym_stride + (size_t)f * kFieldSyms;
#pragma unroll
        for (int ks = 0; ks < kKSteps; ++ks) {
            const uint4 t0 = *reinterpret_cast<const uint4*>(tp + 16 * ks), t1 = *reinterpret_cast<const uint4*>(tp + 16 * ks + 4);
            uint2 sy;
            if constexpr (FUSED) {
                sy = *reinterpret_cast<const uint2*>(ring_rd + ((P0 + 16 * ks) & (kRing - 1)));
            } else {
                const int pos = P0 + 16 * ks + 8 * h;
                const uint2 v = *reinterpret_cast<const uint2*>(sp + (pos < kFieldSyms ? pos : pos - kFieldSyms));
                sy = make_uint2(v.x << kSymShift, v.y << kSymShift);
            }
            make_kstep(sy, t0, t1, bhi[ks], blo[ks]);
        }
    }
    if constexpr (FUSED) {
        lds_order();                                   // the prime has read [P0, P0 + 144)
        produce_until(P0 + (kRound + 4) * kTile + 16);  // what round 0 reads: the newest inputs of tiles tile0+1 .. tile0+9
        lds_order();
    }
    // prefetch FIFOs: entry (t - tile0) % kAhead holds the newest inputs of tile t, for t = tile0 + 1 .. tile0 + kAhead
    uint4 ps[kAhead];
    if constexpr (!FUSED) {
#pragma unroll
        for (int d = 1; d <= kAhead; ++d) ps[d % kAhead] = sym_fetch(tile0 + d);
    }
    // The oscillator windows of a whole round sit in LDS, double-buffered by round parity: slot u holds the 32 newest inputs
    // of tile Tg + u + 1 (what the body of tile Tg + u converts).  The next round's windows are fetched in three groups of
    // three at tiles 0, 3 and 6 of a round and parked three tiles later, INSIDE the same loop iteration: the wait for them is
    // then an exact count (the loads of a group plus the stores issued since), and never drains the wave's IQ stores.  (Loads
    // whose results cross the loop's back edge make hipcc wait for everything at the loop header, and the in-order vmcnt
    // then stalls the wave until its newest stores have landed: 0.3 ms per launch.)
    float ofn[3];
#pragma unroll
    for (int j = 0; j < kRound; ++j) osc_wr[64 * j] = osc_fetch(tile0 + 1 + j);
    f16v zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero16[e] = 0.f;
    f16v acc_prev[UP];
#pragma unroll
    for (int ph = 0; ph < UP; ++ph) acc_prev[ph] = zero16;

    for (int r = 0; r < a.rounds; ++r) {
        const int Tg = tile0 + kRound * r;
        const int obuf = (r & 1) * kRound * 64;   // this round's half of the oscillator windows
        // FUSED: the interleaved bytes of the next two steps of the producer are fetched NOW and used at the round's end (a
        // load issued at production time would wait, in order, behind every IQ store of the round)
        uint32_t pa0 = 0, pa1 = 0, pa2 = 0, pb0 = 0, pb1 = 0, pb2 = 0;
        const bool pre_ok = fs_j < 0;
        if constexpr (FUSED) {
            il_fetch(next_g, pa0, pa1, pa2);
            il_fetch(next_g + 4, pb0, pb1, pb2);
        }
#pragma unroll
        for (int u = 0; u < kRound; ++u) {
            const int tile = Tg + u;
            // Three things run in this tile's body, interleaved by hand so that the vector and LDS work sits between the
            // MFMAs of the chain instead of in front of and behind it (hipcc clusters them otherwise, and the two waves of a
            // SIMD then serialise on the phases):
            //   chain : the 21 MFMAs of THIS tile
            //   out   : the PREVIOUS tile's accumulator -> scale -> line image -> stores
            //   next  : the NEXT tile's two newest K-steps (its 32 newest inputs per column) -> the slots this tile's
            //           K-steps 0 and 1 vacate
            const bool prev_ok = (u > 0 || r > 0) && tile - 1 < kTiles;
            const __amdgpu_buffer_rsrc_t rs_iq = __builtin_amdgcn_make_buffer_rsrc(iq_base, 0, prev_ok ? iq_records : 0, kRsrcFlags);
            const int st_soff = (tile - 1) * G::kRowBytes;
            // chain slots of the tile's side work (UP = 1: the numbers the comments of side() name)
            constexpr int kChain = chain_before(kKSteps) * UP;   // MFMAs per tile
            // (same-box A/B of ten placements -- mix at 3..7, image 4..6 slots behind it, image reads 0..2 slots later: all within
            // 1 %, 1.90-1.92 ms; the 27-MFMA chain with its own best placement: 2.00)
            constexpr int kMixAt = 4 * UP;               // K-steps 0 and 1 have issued: their slots take the next tile's
            constexpr int kImgAt = kMixAt + 4;           // image writes, 4 values per slot
            constexpr int kRdAt = kImgAt + 4 * UP;       // image reads, then the stores two slots behind
            static_assert(kRdAt + 2 + G::kStores <= kChain, "side work fits the chain");
            float y[16 * UP];
            uint2 sy[2];
            uint4 tw[4];
            uint4 piece[G::kStores];
            f16v acc[UP];
#pragma unroll
            for (int ph = 0; ph < UP; ++ph) acc[ph] = zero16;
            auto side = [&](int i) {
                if (i == 0) {
                    if (u % 3 == 0) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) ofn[j] = osc_fetch(Tg + kRound + 1 + u + j);   // three of the next round's windows
                    }
                    if constexpr (!FUSED) {   // next tile's ring piece; the slot's prefetch is re-issued kAhead tiles ahead
                        if (lane < 32) sym_park(tile + 1, ps[(u + 1) % kAhead]);
                        ps[(u + 1) % kAhead] = sym_fetch(tile + 1 + kAhead);
                    }
                }
                if (i >= 1 && i <= 4 * UP) {
                    constexpr float inv = 1.0f / (kAScale * kZScale);
#pragma unroll
                    for (int e = 4 * (i - 1); e < 4 * i; ++e) y[e] = acc_prev[e >> 4][e & 15] * inv;
                }
                if (i == 2) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        sy[s] = *reinterpret_cast<const uint2*>(ring_rd + (((tile + 1) * kTile + 112 + 16 * s) & (kRing - 1)));
                        tw[2 * s] = *reinterpret_cast<const uint4*>(osc_rd + obuf + 64 * u + 16 * s);
                        tw[2 * s + 1] = *reinterpret_cast<const uint4*>(osc_rd + obuf + 64 * u + 16 * s + 4);
                    }
                }
                if (i >= kMixAt && i < kMixAt + 4) {
                    // four elements (one ring dword) of the next tile's newest K-steps, into the slots of this tile's K-steps
                    // 0 and 1, whose MFMAs have issued
                    const int c4 = i - kMixAt, s = c4 >> 1, d = c4 & 1;
                    const int slot = (2 * u + s) % kKSteps;
                    const uint32_t dw = d ? sy[s].y : sy[s].x;
                    const uint4 t = tw[2 * s + d];
                    uint32_t h0, l0, h1, l1;
                    mix_quad(dw, t, h0, l0, h1, l1);
                    if (d == 0) { bhi[slot].x = h0; bhi[slot].y = h1; blo[slot].x = l0; blo[slot].y = l1; }
                    else        { bhi[slot].z = h0; bhi[slot].w = h1; blo[slot].z = l0; blo[slot].w = l1; }
                }
                if (i >= kImgAt && i < kImgAt + 4 * UP) {
                    // C -> line image: register e of lane (n, h) is output row (e & 3) + 8 (e >> 2) + 4 h of column n; with UP = 2
                    // the two phases interleave (sample 2 row + phase)
#pragma unroll
                    for (int e = 4 * (i - kImgAt); e < 4 * (i - kImgAt + 1); ++e)
                        put_component<FMT>(img_wr, UP * ((e & 3) + 8 * ((e & 15) >> 2) + 4 * h) + (e >> 4), comp, y[e]);
                }
                if (u % 3 == 2 && i == kChain - 1) {
                    // the group fetched at tile u - 2 goes to the other buffer; at the round's end the producer's interleaved
                    // bytes are claimed as well, where the wait is still an exact count
#pragma unroll
                    for (int j = 0; j < 3; ++j) osc_wr[(kRound * 64 - obuf) + 64 * (u - 2 + j)] = ofn[j];
                    if constexpr (FUSED) {
                        if (u == kRound - 1) asm volatile("" ::"v"(pa0), "v"(pa1), "v"(pa2), "v"(pb0), "v"(pb1), "v"(pb2));
                    }
                }
                if (i >= kRdAt && i < kRdAt + G::kStores)
                    piece[i - kRdAt] = *reinterpret_cast<const uint4*>(img_rd + (i - kRdAt) * G::kRowsPerStore * G::kImgRow);
                if (i >= kRdAt + 2 && i < kRdAt + 2 + G::kStores) {
                    const int q = i - (kRdAt + 2);
                    const uint4 v = piece[q];
                    // non-temporal: the IQ is written once and never read back by the GPU
                    __builtin_amdgcn_raw_buffer_store_b128((v4u){v.x, v.y, v.z, v.w}, rs_iq, (int)(st_off + q * st_step), st_soff, 2);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
#pragma unroll
            for (int ks = 0; ks < kKSteps; ++ks) {
                const int slot = (2 * u + ks) % kKSteps;
                const h8 bl = __builtin_bit_cast(h8, blo[slot]);
                // where the z correction is left out the product takes z rounded ONCE, fl16(Zhi + Zlo) (four packed adds), instead
                // of Zhi, which carries two roundings (the oscillator's split and the product's): half the error of the omission
                const h8 bh = z_lo_needed(ks) ? __builtin_bit_cast(h8, bhi[slot]) : __builtin_bit_cast(h8, bhi[slot]) + bl;
#pragma unroll
                for (int ph = 0; ph < UP; ++ph) {
                    const h8 ah = ahi[ph][ks], al = alo[ph][ks];
                    // (the slot numbers are written as constants of the unrolled loop variables: with a running counter hipcc
                    // no longer sees side()'s small arrays as registers and moves one of them into LDS, 12 KB and four extra
                    // 16-byte LDS accesses per tile)
                    const int i0 = UP * chain_before(ks) + ph * products_of(ks);
                    acc[ph] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[ph], 0, 0, 0);
                    side(i0);
                    if (tap_lo_needed(ks)) {
                        acc[ph] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[ph], 0, 0, 0);
                        side(i0 + 1);
                    }
                    if (z_lo_needed(ks)) {
                        acc[ph] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[ph], 0, 0, 0);
                        side(i0 + (tap_lo_needed(ks) ? 2 : 1));
                    }
                }
            }
#pragma unroll
            for (int ph = 0; ph < UP; ++ph) acc_prev[ph] = acc[ph];
        }
        if constexpr (FUSED) {
            // the ring moves on to what the next round reads (nothing is needed behind the run's last tile unless the tap wants it)
            if (r + 1 < a.rounds || a.sym) {
                lds_order();
                const int limit = (Tg + 2 * kRound + 4) * kTile + 16;
#pragma unroll 1
                for (int it = 0; it < 2; ++it)
                    if (pre_ok && fs_j < 0 && prod_pos < limit) {
                        produce_quad(it ? pb0 : pa0, it ? pb1 : pa1, it ? pb2 : pa2);
                        tap_blocks(tap_end);
                    }
                produce_until(limit);   // field sync, field end, a third step at a segment boundary
                lds_order();
            }
        }
    }
    {   // the run's last tile: scale, image, stores
        const int tile = tile0 + kRound * a.rounds - 1;
        const __amdgpu_buffer_rsrc_t rs_iq = __builtin_amdgcn_make_buffer_rsrc(iq_base, 0, tile < kTiles ? iq_records : 0, kRsrcFlags);
        constexpr float inv = 1.0f / (kAScale * kZScale);
#pragma unroll
        for (int e = 0; e < 16 * UP; ++e)
            put_component<FMT>(img_wr, UP * ((e & 3) + 8 * ((e & 15) >> 2) + 4 * h) + (e >> 4), comp, acc_prev[e >> 4][e & 15] * inv);
        lds_order();
#pragma unroll
        for (int q = 0; q < G::kStores; ++q) {
            const uint4 v = *reinterpret_cast<const uint4*>(img_rd + q * G::kRowsPerStore * G::kImgRow);
            __builtin_amdgcn_raw_buffer_store_b128((v4u){v.x, v.y, v.z, v.w}, rs_iq, (int)(st_off + q * st_step), tile * G::kRowBytes, 2);
        }
    }
}

void build_mm_operand(const float* taps112, const float* taps_half112, std::vector<uint16_t>& h) {
    h.assign(kMmOperandHalves, 0);
    for (int ph = 0; ph < 2; ++ph)
        for (int ks = 0; ks < kKSteps; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int p = lane & 31, hh = lane >> 5;
                    const int idx = 16 * ks + 8 * hh + j - p;
                    // phase 0: 111 taps; phase 1: 112 taps, centred half a symbol later
                    const float v = ph == 0 ? ((idx >= 0 && idx <= 110) ? taps112[idx] * kAScale : 0.0f)
                                            : ((idx >= 0 && idx <= 111) ? taps_half112[idx] * kAScale : 0.0f);
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    h[((size_t)((2 * ph) * kKSteps + ks) * 64 + lane) * 8 + j] = __builtin_bit_cast(uint16_t, hi);
                    h[((size_t)((2 * ph + 1) * kKSteps + ks) * 64 + lane) * 8 + j] = __builtin_bit_cast(uint16_t, lo);
                }
}

hipError_t upload_mm_tables(const float* taps112, const float* taps_half112) {
    std::vector<uint16_t> h;
    build_mm_operand(taps112, taps_half112, h);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_a_tab), h.data(), sizeof(uint16_t) * h.size());
}

// what the kernels of the current device actually read (tests: the error budget is computed from these bits)
hipError_t download_mm_tables(uint16_t* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_a_tab), sizeof(uint16_t) * kMmOperandHalves);
}

namespace {

// Runs: long runs amortise what a run pays at its start; short runs balance the grid.  The chip holds `resident` of these
// waves at once (2048 = 2 per SIMD; 1024 for the 2x form), so a launch takes about ceil(tasks / resident) waves of (run
// length + start-up) tile times.
int pick_rounds(int ch_groups, int fields, int startup_tiles, int resident) {
    const int max_rounds = (kTiles + kRound - 1) / kRound;   // 905: one run per field
    long best = -1;
    int rounds = max_rounds;
    for (int runs = 1; runs <= 64; ++runs) {
        const int cand = (max_rounds + runs - 1) / runs;
        const long t = (long)((kTiles + kRound * cand - 1) / (kRound * cand)) * ch_groups * fields;
        const long cost = ((t + resident - 1) / resident) * (kRound * cand + startup_tiles);
        if (best < 0 || cost < best) { best = cost; rounds = cand; }
    }
    return rounds;
}

// ev: the dispatch itself carries the two timing events (hipExtLaunchKernelGGL: start and stop come from the kernel's own
// completion signal), so timing this kernel puts no event packets -- about 5 us of idle stream each -- around it
template <int FMT, bool FUSED, int UP>
void launch_mm(hipStream_t s, FirArgs a, const KernelEvents* ev) {
    a.ch_groups = (a.channels + kFusedGroup - 1) / kFusedGroup;
    a.rounds = pick_rounds(a.ch_groups, a.fields, FUSED ? 12 : 4, UP == 1 ? 2048 : 1024);
    a.runs_per_field = (kTiles + kRound * a.rounds - 1) / (kRound * a.rounds);
    const long tasks = (long)a.runs_per_field * a.ch_groups * a.fields;
    const unsigned grid = (unsigned)((tasks + kWaves - 1) / kWaves);
    if (ev)
        hipExtLaunchKernelGGL((k_mix_fir_mm<FMT, FUSED, UP>), dim3(grid), dim3(64 * kWaves), 0, s, ev->start, ev->stop, 0, a);
    else
        hipLaunchKernelGGL((k_mix_fir_mm<FMT, FUSED, UP>), dim3(grid), dim3(64 * kWaves), 0, s, a);
}

template <bool FUSED, int UP>
void launch_fmt(hipStream_t s, const FirArgs& a, int iq_format, const KernelEvents* ev) {
    switch (iq_format) {
        case kIqCs16: launch_mm<kIqCs16, FUSED, UP>(s, a, ev); break;
        case kIqCs8: launch_mm<kIqCs8, FUSED, UP>(s, a, ev); break;
        default: launch_mm<kIqCf32, FUSED, UP>(s, a, ev); break;
    }
}

template <bool FUSED>
void launch_up(hipStream_t s, const FirArgs& a, int iq_format, int upsample, const KernelEvents* ev = nullptr) {
    if (upsample == 2) launch_fmt<FUSED, 2>(s, a, iq_format, ev);
    else launch_fmt<FUSED, 1>(s, a, iq_format, ev);
}

}  // namespace

void launch_mix_fir_mfma(hipStream_t s, const uint8_t* d_sym, size_t sym_stride, void* d_iq, size_t iq_stride,
                         const float* osc_planes, int channels, int fields, int iq_format, int upsample) {
    FirArgs a{};
    a.sym = const_cast<uint8_t*>(d_sym);
    a.sym_stride = sym_stride;
    a.iq = d_iq;
    a.iq_stride = iq_stride;
    a.planes = osc_planes;
    a.channels = channels;
    a.fields = fields;
    launch_up<false>(s, a, iq_format, upsample);
}

void launch_mix_fir_fused(hipStream_t s, const DeviceTables& t, const Buffers& b, uint8_t* d_sym, size_t sym_stride, void* d_iq,
                          size_t iq_stride, const float* osc_planes, int channels, int fields, uint64_t field_base,
                          int iq_format, int upsample, const KernelEvents* ev) {
    FirArgs a{};
    a.sym = d_sym;   // optional symbol tap
    a.sym_stride = sym_stride;
    a.iq = d_iq;
    a.iq_stride = iq_stride;
    a.planes = osc_planes;
    a.channels = channels;
    a.fields = fields;
    a.il = b.il;
    a.il_stride = b.il_stride;
    a.cin = b.cin;
    a.last_group = b.last_group;
    a.fsync = t.fsync;
    a.parity0 = (unsigned)(field_base & 1);
    launch_up<true>(s, a, iq_format, upsample, ev);
}

}  // namespace atsc_k
