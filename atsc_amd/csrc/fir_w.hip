// gfx950 trellis / map / sync + pilot/offset mixer + RRC FIR in one kernel, the oscillator folded into the TAP operand.
//
//   y[c][i] = sum_{k<111} taps[k] * T_f[i+k] * x_c[i+k]              (reference: offset.h:11-45, filter.h:9-69)
//
// k_mix_fir_mm (fir_fused.hip) mixes first (z = x T per channel, two fp16 pieces) and multiplies by a constant Toeplitz tap
// matrix (two fp16 pieces): three matrix products per K-step, 42 MFMAs per 32 channels x 32 outputs, and the board sits on its
// power cap.  Here the roles are cut differently:
//
//   Y[ch][n] = sum_k X[ch][k] * W[k][n],   W[k][2 p + comp] = taps[k - p] * T_comp[i0 + k]     (re: comp 0, im: comp 1)
//
//   * X holds the LEVELS alone (2 v - 7 + 1.25: 6 significant bits), exact in ONE fp16 piece: no low-order product on this side.
//   * W does not depend on the channel: it is regenerated per tile from the taps (per-lane constants, two fp16 pieces) and
//     the batch's oscillator table (two fp16 pieces from k_osc_table) in packed fp16 arithmetic -- W = Whi + Wlo to 2^-21 --
//     and shared by the 64 channels of the wave (two 32-channel row blocks).
//   * N = 32 columns = 16 consecutive outputs x (re, im), K = 128 = eight K-steps of v_mfma_f32_32x32x16_f16 (16 + 110 inputs
//     feed a tile): 8 products with Whi and 6 with Wlo (the taps of K-steps 0 and 7 sum to < 0.00065 per row: their low-order
//     piece is worth 5.5e-6 of full scale at most, scripts/fir_error_budget.py) = 14 MFMAs per 32 channels x 16 outputs,
//     28 per 32 x 32 against 42.
//   * W is the A operand (row 2 p + comp), the levels the B operand (column = channel): a lane of the accumulator tile is a
//     CHANNEL and four consecutive registers are two consecutive samples (re, im, re, im) = 16 contiguous bytes of that
//     channel's row.  They go to a wave-private LDS image with eight ds_write_b128 per tile and leave as WHOLE 128-byte lines,
//     eight dwordx4 stores of eight rows each.  Measured on the way here (scripts/ubench/store_dword_rows.hip,
//     profiles/r03/ubench_store_dword_rows.txt): with the roles the other way round a register is a whole line and the IQ can
//     leave as 32 dword stores per tile straight from the accumulators -- but dword stores beside MFMA work cost 2.1-3.0 ms for
//     this output where dwordx4 stores cost 1.6; dwordx4 stores straight from the accumulators in the present orientation write
//     a line as four 32-byte pieces: twice the write requests between L1 and L2, 2.7 ms.
//   * scales: levels * 2^-12, taps * 2^8, oscillator * 2^4 -- every piece in fp16's normal range and the product at its true
//     scale, so the accumulators ARE the samples (no post-scaling).  (v_mfma_f32_32x32x16_f16 honours fp16 subnormals on both
//     operands: scripts/ubench/mfma_denorm.hip.)
//
// One wave = one (field, run of tiles, group of 64 channels) task, two waves per SIMD (229 VGPRs, 78 KB of LDS per four-wave
// workgroup).  Symbols reach the wave through a wave-private LDS ring [64 channels][128 positions] filled by its own precoder /
// trellis / mapper / sync producer from the interleaved bytes (reference: trellis.h:10-51, field_sync.h:12-26; lane = channel,
// one byte-time = 48 symbols per step, a step every third tile or so); the oscillator through a wave-private LDS ring of 256
// positions per component.  No workgroup barriers.
//
// Where it stands (64 channels x 64 fields): 1.99 ms per launch against k_mix_fir_mm's 1.86 -- a third fewer matrix
// products, but the per-tile W (104 packed fp16 instructions) and the producer of 64 channels cost the SIMD as many issue
// cycles as the products saved; the board draws 1314 W at 2.3 GHz here against 1400 W (its cap) at 1.9-2.0 GHz there.  Its
// measured error is the lowest of the matrix-core forms (1.2e-6 of full scale against 1.7e-6); its worst case with every
// term at its bound is not (8.4e-6 against 7.2e-6, scripts/fir_error_budget.py).  Kept as the measured alternative
// (ATSC_HIP_FIR_WIDE); AUTO stays with k_mix_fir_mm.
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
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kWT = 16;                             // outputs per tile
constexpr int kWK = 8;                              // K-steps per tile: 128 >= 16 + 110 inputs
constexpr int kWTiles = kFieldSyms / kWT;           // 16276 tiles per field
constexpr int kWRound = 8;                          // tiles per unrolled round = period of the K-step slot renaming
constexpr int kWWaves = 4;
constexpr int kWRing = 128;                         // positions per channel row of the symbol ring (the producer stays <= 100 ahead of the tile)
constexpr int kWRingRow = kWRing + 8;               // bytes: 34 dwords = 2 x odd -> the 32 rows' b64 reads of a half-wave are conflict-free
constexpr int kWOscRing = 256;                      // positions per component of the oscillator ring: a tile's 128 inputs + a round's 128 new ones
constexpr int kWByteTimes = kFieldCoded / 12;       // 5382 byte-times per field, 207 per part
constexpr int kWImgRow = kWT * 8 + 16;              // bytes: 36 dwords -> the 16 lanes of a b128 service group (rows c .. c + 15, same column) hit 64 different banks
constexpr int kWSymShift = 5;                       // ring bytes are v << 5: fp16 (0x3c00 | byte) = 1 + v / 32
constexpr float kWTapScale = 256.0f;                // taps * 2^8 (the smallest, 4e-6, stays a normal fp16 number)

static_assert(kFieldSyms % kWT == 0, "tiles must cover a field exactly");
static_assert(kWT + 110 <= 16 * kWK, "K must cover the filter");
static_assert(kWRound == kWK, "a tile advances by one K-step: the slot renaming has period kWK");
static_assert(kWideGroup == 64, "one wave task = 64 channels");


// tap constants in the B-operand lane layout: [hi | lo][K-step][lane] x 16 bytes = the four element pairs of the lane
__device__ uint4 g_w_tab[2 * kWK * 64];

struct WArgs {
    uint8_t* sym;              // optional symbol tap
    size_t sym_stride;
    void* iq;
    size_t iq_stride;          // samples per channel row
    const float* planes;       // [field][re plane | im plane] of kPaddedSyms dwords (k_osc_table, planes form, scale 2^4)
    int channels, fields, rounds, runs_per_field, ch_groups;
    const uint8_t* il;
    size_t il_stride;
    const uint4* cin;
    const uint8_t* last_group;
    const uint8_t* fsync;
    unsigned parity0;
};

__device__ __forceinline__ void lds_order_w() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Eight ring bytes -> eight fp16 levels (2 v - 5.75) * 2^-12 (signal.h:13-18): 0x3c00 | (v << 5) is 1 + v / 32; times 2^-6
// minus 279 * 2^-14 is (8 v - 23) * 2^-14, exact.
__device__ __forceinline__ uint4 levels8(uint2 ring) {
    const h2 k1 = {(_Float16)0.015625f, (_Float16)0.015625f};
    const h2 k2 = {(_Float16)(-279.0f / 16384.0f), (_Float16)(-279.0f / 16384.0f)};
    auto two = [&](uint32_t dw, uint32_t sel) {
        const uint32_t pr = __builtin_amdgcn_perm(0x3c3c3c3cu, dw, sel);
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(h2, pr), k1, k2));
    };
    return make_uint4(two(ring.x, 0x04010400u), two(ring.x, 0x04030402u), two(ring.y, 0x04010400u), two(ring.y, 0x04030402u));
}

__device__ __forceinline__ h2 as_h2(uint32_t v) { return __builtin_bit_cast(h2, v); }
__device__ __forceinline__ uint32_t as_u32(h2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t& comp_of(uint4& v, int q) { return q == 0 ? v.x : (q == 1 ? v.y : (q == 2 ? v.z : v.w)); }
__device__ __forceinline__ uint32_t comp_of(const uint4& v, int q) { return q == 0 ? v.x : (q == 1 ? v.y : (q == 2 ? v.z : v.w)); }

// W = taps * T in two fp16 pieces, the four element pairs of a lane's K-step at once and STAGE by stage (a packed fp16
// instruction that reads the result of the one just before it costs a wait state):
//   stage 0: whi = fl16(ah th)                         stage 1: r = ah th - whi   (exact: the rounding error of an fp16 product is an fp16 number)
//   stage 2: l = fl16(ah tl + r)                       stage 3: wlo = fl16(al th + l)
// t holds {th, tl} of pairs 0, 1 in t[0].xy / .zw and of pairs 2, 3 in t[1]
__device__ __forceinline__ void w_stage(int stage, const uint4& ah, const uint4& al, const uint4 (&t)[2], uint4& whi, uint4& wlo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t th = (q & 1) ? t[q >> 1].z : t[q >> 1].x, tl = (q & 1) ? t[q >> 1].w : t[q >> 1].y;
        const h2 a = as_h2(comp_of(ah, q));
        if (stage == 0) comp_of(whi, q) = as_u32(a * as_h2(th));
        else if (stage == 1) comp_of(wlo, q) = as_u32(__builtin_elementwise_fma(a, as_h2(th), -as_h2(comp_of(whi, q))));
        else if (stage == 2) comp_of(wlo, q) = as_u32(__builtin_elementwise_fma(a, as_h2(tl), as_h2(comp_of(wlo, q))));
        else comp_of(wlo, q) = as_u32(__builtin_elementwise_fma(as_h2(comp_of(al, q)), as_h2(th), as_h2(comp_of(wlo, q))));
    }
}

constexpr bool w_lo_needed(int ks) { return ks >= 1 && ks <= 6; }
constexpr int w_mfmas(int ks) { return w_lo_needed(ks) ? 4 : 2; }      // per K-step: two row blocks x (hi [+ lo])
constexpr int w_slot0(int ks) { return ks == 0 ? 0 : 2 + 4 * (ks - 1); }   // chain slot of the K-step's first MFMA
constexpr int kWChain = 28;                                             // MFMAs per tile
static_assert(w_slot0(7) + w_mfmas(7) == kWChain, "28 MFMAs per tile");

}  // namespace

__global__ __launch_bounds__(64 * kWWaves, 2) void k_mix_fir_w(const WArgs a) {
    __shared__ __align__(16) uint8_t s_ring[kWWaves][kWideGroup * kWRingRow];
    __shared__ __align__(16) uint8_t s_img[kWWaves][kWideGroup * kWImgRow];   // the finished tile: [channel row][16 samples x 8 B], rows 144 B apart
    __shared__ __align__(16) uint32_t s_osc[kWWaves][2 * kWOscRing];   // [re | im][position & 255]: {Th pair}, {Tl pair} dwords as in the planes

    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, h = lane >> 5;
    const long task = (long)blockIdx.x * kWWaves + wv;
    const int run = (int)(task % a.runs_per_field);
    const long gidx = task / a.runs_per_field;
    const int cg = (int)(gidx % a.ch_groups);
    const int f = (int)(gidx / a.ch_groups);
    if (f >= a.fields) return;
    const int tile0 = run * a.rounds * kWRound;
    const int P0 = tile0 * kWT;
    constexpr int rows_valid = kWideGroup;   // whole groups only (checked on the host): the IQ stores carry no range check

    // ---- tap constants of this lane (B-operand layout: column n = 2 p + comp, k = 8 h + j)
    uint4 ah[kWK], al[kWK];
#pragma unroll
    for (int ks = 0; ks < kWK; ++ks) {
        ah[ks] = g_w_tab[ks * 64 + lane];
        al[ks] = g_w_tab[(kWK + ks) * 64 + lane];
    }

    // ---- IQ out: lane (n, h) of row block b holds channel row 32 b + n; its registers 4 g .. 4 g + 3 are the floats
    // 8 g + 4 h .. + 3 of the tile's 32 (16 samples x re, im) = bytes 32 g + 16 h of the row's 128-byte line.  They go to a
    // wave-private LDS image as 16-byte pieces and come back as WHOLE lines (lane = row 8 q + lane / 8, piece lane % 8), which
    // leave as eight dwordx4 stores of eight rows each through buffer descriptors (32-bit offsets: eight rows of a batch stay
    // below 4 GiB, checked on the host; tiles that do not exist are dropped by the range check).  Stored straight from the
    // accumulators a line would leave as four 32-byte pieces: twice the write requests between L1 and L2, 2.7 ms instead of 1.6.
    constexpr int kRsrcFlags = 0x00020000;
    const uint32_t row_bytes = (uint32_t)(a.iq_stride * sizeof(float2));
    uint8_t* iq_base = reinterpret_cast<uint8_t*>(a.iq) + ((size_t)cg * kWideGroup * a.iq_stride + (size_t)f * kFieldSyms) * sizeof(float2);
    const uint32_t st_off = (uint32_t)(lane >> 3) * row_bytes + 16u * (lane & 7);
    uint8_t* img = s_img[wv];
    uint8_t* img_wr = img + n * kWImgRow + 16 * h;                        // + 32 b rows, + 32 g bytes
    const uint8_t* img_rd = img + (lane >> 3) * kWImgRow + 16 * (lane & 7);   // + 8 q rows

    const __amdgpu_buffer_rsrc_t rs_osc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.planes) + (size_t)f * 2 * kPaddedSyms, 0, 2 * kPaddedSyms * 4, kRsrcFlags);
    const int osc_off = (h * kPaddedSyms + n) * 4;   // lanes 0..31 the re plane, 32..63 the im plane; reads past the table return 0
    auto osc_fetch = [&](int pos) {
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_osc, osc_off, pos * 4, 0);
    };
    uint32_t* oscw = s_osc[wv];
    uint32_t* osc_wr = oscw + h * kWOscRing;                         // + ((position + n) & 255)
    const uint32_t* osc_rd = oscw + (n & 1) * kWOscRing + 8 * h;     // + ((position of the K-step) & 255): this lane's 8 inputs

    uint8_t* ring = s_ring[wv];
    const uint8_t* ring_rd0 = ring + n * kWRingRow + 8 * h;              // row block 0: channel n; + (position & 255)
    const uint8_t* ring_rd1 = ring_rd0 + 32 * kWRingRow;                 // row block 1: channel 32 + n

    // ---- the symbol producer: lane = channel of the group.  One step turns one byte-time (12 interleaved bytes = 48 data
    // symbols) of the 64 channels into ring symbols -- precoder, trellis encoder, mapper, segment syncs exactly as
    // k_trellis_map (kernels.hip; reference trellis.h:10-51, field_sync.h:12-26); the trellis state stays in the lane.
    const int pch = cg * kWideGroup + min(lane, rows_valid - 1);          // rows past the end repeat the last one; never stored
    const uint8_t* il_c = a.il + (size_t)pch * a.il_stride + (size_t)f * kFieldCoded;
    uint8_t* pring = ring + lane * kWRingRow;
    int next_g = 0;        // next byte-time of the field to produce (uniform)
    int fs_j = -1;         // next field-sync symbol to produce, or -1 when not inside the field-sync segment
    int fs_base = 0;       // 0: the field's own sync segment; kFieldSyms: its copy behind the field (field_sync.h:16)
    int prod_pos = 0;      // every (extended) position below it is in the ring
    int tap_pos = 0;       // next 128-symbol block of the symbol tap
    uint32_t cw[3] = {0, 0, 0}, rsv[3] = {0, 0, 0};
    const uint32_t* fs_tab = reinterpret_cast<const uint32_t*>(a.fsync + (((a.parity0 + (unsigned)f) & 1u) ? kSegSyms : 0));
    __amdgpu_buffer_rsrc_t rs_tap = rs_osc;
    if (a.sym)
        rs_tap = __builtin_amdgcn_make_buffer_rsrc(a.sym + (size_t)cg * kWideGroup * a.sym_stride + (size_t)f * kFieldSyms, 0,
                                                   (int)((uint32_t)rows_valid * (uint32_t)a.sym_stride), kRsrcFlags);
    {
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
        tap_pos = (P0 + 63) & ~63;
        // Trellis state entering next_g: the part's carry and the parities of the part's earlier byte-times.  The parity is
        // GF(2)-linear in the data bits, so the canonical words are XORed first and the parity is taken once.
        {
            uint32_t ex[3] = {0, 0, 0};
            // sixteen byte-times (192 contiguous bytes per lane, twelve 16-byte loads at dword alignment) in flight: the loop is
            // latency-bound, and as 48 dword loads per step it cost 0.115 ms per launch
            typedef uint32_t u4a __attribute__((ext_vector_type(4), aligned(4)));
            for (int g0 = part0 * kPartGroups; g0 < next_g; g0 += 16) {
                uint32_t d[48];
#pragma unroll
                for (int k4 = 0; k4 < 12; ++k4) {
                    // (a step may read up to 180 bytes past the run's start: still inside the field's interleaved bytes or the
                    // next field's / the buffer's 256-byte row padding; the words are masked out below)
                    const u4a v = *reinterpret_cast<const u4a*>(il_c + 12 * (size_t)min(g0, kWByteTimes - 16) + 16 * k4);
                    d[4 * k4] = v.x; d[4 * k4 + 1] = v.y; d[4 * k4 + 2] = v.z; d[4 * k4 + 3] = v.w;
                }
                const int gb = min(g0, kWByteTimes - 16);   // the step's first byte-time as loaded
#pragma unroll
                for (int k16 = 0; k16 < 16; ++k16) {
                    const int g = gb + k16;
                    if (g >= g0 && g < next_g) {
                        const Words3 e = canonical_group(&d[3 * k16], g);
#pragma unroll
                        for (int k = 0; k < 3; ++k) ex[k] ^= e.w[k];
                    }
                }
            }
            const uint4 ci = cin_c[part0];
            const uint32_t civ[3] = {ci.x, ci.y, ci.z};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                uint32_t z2, z0, pr;
                trellis_local(ex[k], z2, z0, pr);
                cw[k] = civ[k] ^ pr;
            }
        }
        // reserved symbols of the field-sync segment (field_sync.h:17,23): the last 12 data symbols of the previous field
        {
            const uint4 c0 = cin_c[0];
            const uint32_t cinw[3] = {c0.x, c0.y, c0.z};
            const uint32_t* lg = (f == 0) ? reinterpret_cast<const uint32_t*>(a.last_group + (size_t)pch * 16)
                                          : reinterpret_cast<const uint32_t*>(il_c - 12);
            const Words3 pe = canonical_group(lg, kWByteTimes - 1);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                rsv[(k + 1) % 3] = ((cinw[k] & 0x01010101u) << 2) | ((pe.w[k] & 0x01010101u) << 1) | ((cinw[k] >> 2) & 0x01010101u);
        }
    }
    auto il_fetch = [&](int g, uint32_t& d0, uint32_t& d1, uint32_t& d2) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(il_c) + 3 * min(g, kWByteTimes - 1);
        d0 = src[0]; d1 = src[1]; d2 = src[2];
    };
    // one byte-time: next_g (< kWByteTimes), whose three interleaved dwords are d0..d2
    auto produce_one = [&](uint32_t d0, uint32_t d1, uint32_t d2) {
        const int gg = next_g;
        const uint32_t d3[3] = {d0, d1, d2};
        const Words3 e = canonical_group(d3, gg);
        uint32_t hi[3], zz0[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            uint32_t z2, z0, pr;
            trellis_local(e.w[k], z2, z0, pr);
            const uint32_t cwk = cw[k];
            cw[k] ^= pr;
            const uint32_t zz2 = z2 ^ ((cwk & 0x01010101u) * 0xAAu);
            zz0[k] = z0 ^ (((cwk >> 1) & 0x01010101u) * 0x44u) ^ (((cwk >> 2) & 0x01010101u) * 0x11u);
            hi[k] = zz2 | (e.w[k] & 0x55555555u);
        }
        // the four symbol times of the byte: segment, place in the segment and commutator phase once per symbol time, then
        // the three 4-encoder words go to slots (k - phase) mod 3; ring bytes are v << 5: (Z2 Z1) = bits 7 - 2 tau, 6 - 2 tau
        // of hi -> bits 7, 6; Z0 = bit 6 - 2 tau of zz0 -> bit 5
        const int pp = gg / kPartGroups, tq = gg - pp * kPartGroups;
        const int t0 = 4 * tq;
        int sl = t0 / 69, q = t0 - 69 * sl;
        int m = sl % 3;
#pragma unroll
        for (int tau = 0; tau < 4; ++tau) {
            const int seg = 1 + 12 * pp + sl;
            const int base = kSegSyms * seg + 4 + 12 * q;
            if (q == 0) *reinterpret_cast<uint32_t*>(pring + ((kSegSyms * seg) & (kWRing - 1))) = 0x06010106u << kWSymShift;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t z = tau == 0 ? (zz0[k] >> 1) : (zz0[k] << (2 * tau - 1));
                const uint32_t v = ((hi[k] << (2 * tau)) & 0xC0C0C0C0u) | (z & 0x20202020u);
                int hslot = k - m;
                hslot += hslot < 0 ? 3 : 0;
                *reinterpret_cast<uint32_t*>(pring + ((base + 4 * hslot) & (kWRing - 1))) = v;
            }
            if (++q == 69) { q = 0; ++sl; m = m == 2 ? 0 : m + 1; }
        }
        next_g += 1;
        if (next_g >= kWByteTimes) {
            fs_j = 0; fs_base = kFieldSyms; prod_pos = kFieldSyms;    // behind the field: the copy of its sync head
        } else {
            // first data symbol of byte-time next_g, less the segment sync that may precede it (the state after the fourth symbol time)
            prod_pos = kSegSyms * (1 + 12 * pp + sl) + 12 * q;
        }
    };
    // symbol tap: every complete 64-symbol block below prod_pos -> d_sym, 64-byte pieces, four stores per block (a test path)
    const int tap_end = P0 + kWRound * kWT * a.rounds - 1;   // the run taps the blocks that start inside it
    auto tap_blocks = [&](int upto) {
        if (!a.sym) return;
        const int end = min(min(prod_pos, upto + 63), kFieldSyms + 63) & ~63;
        while (tap_pos < end) {
            const int pp = tap_pos + 16 * (lane & 3);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 16 * q + (lane >> 2);
                const uint8_t* src = ring + row * kWRingRow + (pp & (kWRing - 1));
                const uint2 x = *reinterpret_cast<const uint2*>(src), y = *reinterpret_cast<const uint2*>(src + 8);
                const uint32_t off = pp < kFieldSyms ? (uint32_t)row * (uint32_t)a.sym_stride + (uint32_t)pp : 0xffffffffu;
                __builtin_amdgcn_raw_buffer_store_b128((v4u){x.x >> kWSymShift, x.y >> kWSymShift, y.x >> kWSymShift, y.y >> kWSymShift},
                                                       rs_tap, (int)off, 0, 2);
            }
            tap_pos += 64;
        }
    };
    // everything below `limit` (extended position) into the ring
    auto produce_until = [&](int limit) {
        while (prod_pos < limit) {
            if (fs_j >= 0) {
                // field-sync segment: the table, its last three dwords replaced by the reserved symbols of the lane's channel
                const int nsym = min(min(kSegSyms - fs_j, ((limit - prod_pos) + 3) & ~3), 48);   // (the tap keeps up in steps)
                for (int d = fs_j >> 2; d < ((fs_j + nsym) >> 2); ++d) {
                    const uint32_t v = d < 205 ? fs_tab[d] : (d == 205 ? rsv[0] : (d == 206 ? rsv[1] : rsv[2]));
                    *reinterpret_cast<uint32_t*>(pring + ((fs_base + 4 * d) & (kWRing - 1))) = v << kWSymShift;
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
            produce_one(d0, d1, d2);
            tap_blocks(tap_end);
        }
    };

    // ---- prime: the eight K-steps of the run's first tile from the symbols of [P0, P0 + 128), then the ring moves on to
    // [P0 + 128, P0 + 256), the newest K-steps of round 0's tiles; oscillator of [P0, P0 + 256)
    uint4 lv[2][kWK];
#pragma unroll
    for (int i = 0; i < 8; ++i) osc_wr[(P0 + n + 32 * i) & (kWOscRing - 1)] = osc_fetch(P0 + 32 * i);
#pragma unroll
    for (int half = 0; half < 2; ++half) {   // (the ring holds 128 positions and a producer step may run 52 past its limit)
        produce_until(P0 + 64 * (half + 1));
        lds_order_w();
#pragma unroll
        for (int ks = 4 * half; ks < 4 * half + 4; ++ks) {
            lv[0][ks] = levels8(*reinterpret_cast<const uint2*>(ring_rd0 + ((P0 + 16 * ks) & (kWRing - 1))));
            lv[1][ks] = levels8(*reinterpret_cast<const uint2*>(ring_rd1 + ((P0 + 16 * ks) & (kWRing - 1))));
        }
        lds_order_w();
    }
    produce_until(P0 + 144);
    lds_order_w();
    uint32_t pa[3];
    int pa_g = next_g;
    il_fetch(next_g, pa[0], pa[1], pa[2]);
    // W pipeline: tq[s & 1] holds the oscillator pieces of K-step sequence number s (tile t, K-step ks: inputs 16 (t + ks) + 8 h ..),
    // whi / wlo[s & 1] its W pieces; during the MFMAs of s the pieces of s + 2 are read and W of s + 1 is formed.
    uint4 tq[2][2], whi[2], wlo[2];
    auto t_read = [&](int pos, uint4 (&dst)[2]) {
        const uint32_t* p = osc_rd + (pos & (kWOscRing - 1));
        dst[0] = *reinterpret_cast<const uint4*>(p);
        dst[1] = *reinterpret_cast<const uint4*>(p + 4);
    };
    t_read(P0, tq[0]);
    t_read(P0 + 16, tq[1]);
    wlo[0] = wlo[1] = whi[1] = make_uint4(0, 0, 0, 0);
    w_stage(0, ah[0], al[0], tq[0], whi[0], wlo[0]);
    f16v zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero16[e] = 0.f;
    f16v acc[2];
    acc[0] = acc[1] = zero16;
    for (int r = 0; r < a.rounds; ++r) {
        const int Tg = tile0 + kWRound * r;
        // fetched NOW, used at the round's end (a load issued there would wait, in order, behind every IQ store of the round):
        // the oscillator of the next round's 128 new inputs
        uint32_t ofn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ofn[i] = osc_fetch(P0 + 256 + 128 * r + 32 * i);
#pragma unroll
        for (int u = 0; u < kWRound; ++u) {
            const int tile = Tg + u;
            // Three things run in this tile's body, interleaved by hand between the MFMAs:
            //   out   : the PREVIOUS tile's accumulators -> 32 dword stores straight from the registers, row block b in front of
            //           this tile's first MFMA into block b (one accumulator set: the register budget of two waves per SIMD)
            //   chain : the 28 MFMAs of THIS tile
            //   next  : W of the next K-step, and the NEXT tile's newest K-step of levels -> the slot this tile's K-step 0 vacates
            const bool prev_ok = (u > 0 || r > 0) && tile - 1 < kWTiles;
            uint2 sy[2];
            auto to_image = [&](int b) {   // the previous tile's row block b -> the line image (in front of this tile's first MFMA into b)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<v4f*>(img_wr + 32 * b * kWImgRow + 32 * g) =
                        (v4f){acc[b][4 * g], acc[b][4 * g + 1], acc[b][4 * g + 2], acc[b][4 * g + 3]};
            };
            __amdgpu_buffer_rsrc_t rs_iq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                rs_iq[q] = __builtin_amdgcn_make_buffer_rsrc(iq_base + (size_t)(8 * q) * row_bytes, 0, prev_ok ? (int)(8u * row_bytes) : 0, kRsrcFlags);
            const int st_soff = (tile - 1) * (kWT * (int)sizeof(float2));
            uint4 piece[2];
            auto side = [&](int ks, int j) {
                const int i = w_slot0(ks) + j;               // chain slot
                const int s1 = (ks + 1) & 1;                 // buffers of the next K-step in sequence
                const int ksn = (ks + 1) & (kWK - 1);        // its tap K-step
                // oscillator pieces of the K-step after the next: tile + (ks + 2) / 8, K-step (ks + 2) % 8 -> inputs 16 (tile + ks + 2) or,
                // behind the tile's end, 16 (tile + 1 + (ks + 2 - 8))
                if (j == 0) {
                    const int ks2 = ks + 2;
                    const int pos = ks2 < kWK ? (tile + ks2) * kWT : (tile + 1 + ks2 - kWK) * kWT;
                    t_read(pos, tq[ks & 1]);
                }
                // W of the next K-step: one stage of its four pairs per slot (two stages per slot in the two-MFMA K-steps)
                if (w_mfmas(ks) == 4) {
                    if (j == 0 || w_lo_needed(ksn)) w_stage(j, ah[ksn], al[ksn], tq[s1], whi[s1], wlo[s1]);
                } else {
                    if (j == 0) w_stage(0, ah[ksn], al[ksn], tq[s1], whi[s1], wlo[s1]);
                    if (w_lo_needed(ksn)) {
                        w_stage(2 * j + (j == 0 ? 1 : 0), ah[ksn], al[ksn], tq[s1], whi[s1], wlo[s1]);
                        if (j == 1) w_stage(3, ah[ksn], al[ksn], tq[s1], whi[s1], wlo[s1]);
                    }
                }
                // the next tile's newest K-step of levels: inputs 16 (tile + 1) + 112 .. + 127, into the slot of this tile's K-step 0
                if (i == 4) {
                    sy[0] = *reinterpret_cast<const uint2*>(ring_rd0 + (((tile + 1) * kWT + 112) & (kWRing - 1)));
                    sy[1] = *reinterpret_cast<const uint2*>(ring_rd1 + (((tile + 1) * kWT + 112) & (kWRing - 1)));
                }
                if (i == 8) lv[0][u] = levels8(sy[0]);
                if (i == 10) lv[1][u] = levels8(sy[1]);
                // the previous tile's lines: read from the image at slots 3, 5, .. 17, stored two slots later (non-temporal: the
                // IQ is written once and never read back by the GPU)
                if (i >= 3 && i <= 17 && (i & 1)) {
                    const int q = (i - 3) >> 1;
                    piece[q & 1] = *reinterpret_cast<const uint4*>(img_rd + 8 * q * kWImgRow);
                }
                if (i >= 5 && i <= 19 && (i & 1)) {
                    const int q = (i - 5) >> 1;
                    const uint4 v = piece[q & 1];
                    __builtin_amdgcn_raw_buffer_store_b128((v4u){v.x, v.y, v.z, v.w}, rs_iq[q], (int)st_off, st_soff, 2);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
#pragma unroll
            for (int ks = 0; ks < kWK; ++ks) {
                const int slot = (u + ks) & (kWK - 1);
                const h8 x0 = __builtin_bit_cast(h8, lv[0][slot]), x1 = __builtin_bit_cast(h8, lv[1][slot]);
                const h8 bh = __builtin_bit_cast(h8, whi[ks & 1]), bl = __builtin_bit_cast(h8, wlo[ks & 1]);
                if (ks == 0) { to_image(0); __builtin_amdgcn_sched_barrier(0); }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, x0, ks == 0 ? zero16 : acc[0], 0, 0, 0);
                side(ks, 0);
                if (ks == 0) { to_image(1); lds_order_w(); __builtin_amdgcn_sched_barrier(0); }
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, x1, ks == 0 ? zero16 : acc[1], 0, 0, 0);
                side(ks, 1);
                if (w_lo_needed(ks)) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, x0, acc[0], 0, 0, 0);
                    side(ks, 2);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, x1, acc[1], 0, 0, 0);
                    side(ks, 3);
                }
            }
            // ---- the symbol ring stays 144 positions ahead of the tile: one byte-time (48 symbols) every third tile or so, from
            // the interleaved bytes fetched when the previous one was produced (a tile time ago: the wait is short and counts
            // only the few stores issued since); field sync, field end and the run's first steps take the general path
            const int lim = P0 + 128 * r + 16 * (u + 1) + 144;
            if (prod_pos < lim && (r + 1 < a.rounds || u + 1 < kWRound || a.sym)) {
                lds_order_w();
                if (fs_j < 0 && pa_g == next_g && next_g < kWByteTimes) {
                    produce_one(pa[0], pa[1], pa[2]);
                    tap_blocks(tap_end);
                }
                produce_until(lim);
                pa_g = next_g;
                il_fetch(next_g, pa[0], pa[1], pa[2]);
                lds_order_w();
            }
        }
        // ---- the round's end: the oscillator ring moves on to what the next round reads
        lds_order_w();
#pragma unroll
        for (int i = 0; i < 4; ++i) osc_wr[(P0 + 256 + 128 * r + n + 32 * i) & (kWOscRing - 1)] = ofn[i];
        lds_order_w();
    }
    {   // the run's last tile
        const int tile = tile0 + kWRound * a.rounds - 1;
        const bool ok = tile < kWTiles;
        lds_order_w();
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f16v& c = acc[b];
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<v4f*>(img_wr + 32 * b * kWImgRow + 32 * g) = (v4f){c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
        }
        lds_order_w();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint4 v = *reinterpret_cast<const uint4*>(img_rd + 8 * q * kWImgRow);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(iq_base + (size_t)(8 * q) * row_bytes, 0, ok ? (int)(8u * row_bytes) : 0, kRsrcFlags);
            __builtin_amdgcn_raw_buffer_store_b128((v4u){v.x, v.y, v.z, v.w}, rs, (int)st_off, tile * (kWT * (int)sizeof(float2)), 2);
        }
    }
}


void build_w_operand(const float* taps112, std::vector<uint16_t>& hst) {
    hst.assign(kWOperandHalves, 0);
    for (int ks = 0; ks < kWK; ++ks)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int p = (lane & 31) >> 1, hh = lane >> 5;
                const int idx = 16 * ks + 8 * hh + j - p;
                const float v = (idx >= 0 && idx <= 110) ? taps112[idx] * kWTapScale : 0.0f;
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                hst[((size_t)ks * 64 + lane) * 8 + j] = __builtin_bit_cast(uint16_t, hi);
                hst[((size_t)(kWK + ks) * 64 + lane) * 8 + j] = __builtin_bit_cast(uint16_t, lo);
            }
}

hipError_t upload_w_tables(const float* taps112) {
    std::vector<uint16_t> hst;
    build_w_operand(taps112, hst);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_w_tab), hst.data(), sizeof(uint16_t) * hst.size());
}

hipError_t download_w_tables(uint16_t* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w_tab), sizeof(uint16_t) * kWOperandHalves);
}

void launch_mix_fir_wide(hipStream_t s, const DeviceTables& t, const Buffers& b, uint8_t* d_sym, size_t sym_stride, void* d_iq,
                         size_t iq_stride, const float* osc_planes, int channels, int fields, uint64_t field_base) {
    WArgs a{};
    a.sym = d_sym;
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
    a.ch_groups = (channels + kWideGroup - 1) / kWideGroup;
    // Runs: long runs amortise what a run pays at its start, short runs balance the grid; the chip holds 2048 of these waves
    // at once (two per SIMD), so a launch takes about ceil(tasks / 2048) waves of (run length + start-up) tile times.
    const int max_rounds = (kWTiles + kWRound - 1) / kWRound;   // 2035: one run per field
    long best = -1;
    int rounds = max_rounds;
    for (int runs = 1; runs <= 128; ++runs) {
        const int cand = (max_rounds + runs - 1) / runs;
        const long tasks = (long)((max_rounds + cand - 1) / cand) * a.ch_groups * fields;
        const long cost = ((tasks + 2047) / 2048) * ((long)kWRound * cand + 16);
        if (best < 0 || cost < best) { best = cost; rounds = cand; }
    }
    a.rounds = rounds;
    a.runs_per_field = (max_rounds + rounds - 1) / rounds;
    const long tasks = (long)a.runs_per_field * a.ch_groups * fields;
    const unsigned grid = (unsigned)((tasks + kWWaves - 1) / kWWaves);
    hipLaunchKernelGGL(k_mix_fir_w, dim3(grid), dim3(64 * kWWaves), 0, s, a);
}

}  // namespace atsc_k
