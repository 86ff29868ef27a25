// gfx950 mixer + RRC FIR for MANY channels as a banded GEMM on the matrix cores.
//
//   y[c][i] = sum_{k<111} taps[k] * T_f[i+k] * x_c[i+k]          (reference: offset.h:11-45, filter.h:9-69)
//
// All channels of a handle share the oscillator T_f (offset.h:9: one table per encoder, advanced once per field), so
// for a tile of 16 consecutive outputs i0..i0+15 the products W[p][m] = taps[m-p] * T_f[i0+m] (m < 126) are the same for
// every channel and
//
//   Y[2p+comp][c] = sum_m W_comp[p][m] * X[m][c],        X[m][c] = level(symbol of channel c at i0+m)
//
// is a GEMM: M = 32 rows (16 outputs x re/im), N = channels, K = 128.  The levels 2v - 5.75 are EXACT in fp16, and a
// weight splits into two fp16 pieces hi + lo (scaled by 2^16 so that both stay in fp16's range) that carry 21+ bits of
// it, so the products are exact and the fp32 accumulation of v_mfma_f32_32x32x16_f16 gives the same accuracy class as
// the fp32 paths (measured max error vs the reference: see DESIGN.md) at 16x the fp32 vector rate.
//
// One wave = one (field, run of tiles, group of 32*NT channels) task:
//   * A operand (weights) is generated in registers, straight in the MFMA lane layout (lane = row 2p+comp, 8 consecutive
//     m): 8 taps from LDS (four copies of the tap array shifted by 0..3 so that every lane's window is 16-byte aligned),
//     8 oscillator values from a wave-private LDS window (the 256 re and 256 im values that 8 consecutive tiles need,
//     refilled from the batch table's re / im planes once per 8 tiles, the next refill prefetched in registers),
//     multiply, split -- shared by the NT channel tiles of the wave.
//   * B operand (levels) is a sliding window of 8 K-steps in registers: a tile advances by 16 outputs = exactly one
//     K-step, so each tile converts only 8 new symbols per lane (bytes -> fp16 by an exponent trick, exact for every
//     byte value) and the rest of the window is a renaming of registers (the tile loop is unrolled by 8).  The symbol
//     bytes come from a wave-private LDS window that is refilled once per 8 tiles with whole 128-byte pieces of every
//     channel's row (coalesced 16-byte loads, each symbol line fetched once): per-tile 8-byte loads would touch 32
//     lines per instruction, and with the IQ write stream flushing the L2 those lines were re-fetched from HBM for
//     every tile (measured: the kernel ran at 0.4 of its own no-store speed).
//   * C (32x32 fp32) has re/im of two consecutive outputs in four consecutive registers of a lane: 16 bytes of the IQ
//     stream.  It leaves through a wave-private LDS image [channel][16 outputs] so that every global store instruction
//     writes whole 128-byte lines (8 channels x one line), in the handle's sample format.
// No workgroup barriers after the tap table is staged; waves are independent (LDS operations of a wave execute in
// program order).
#include "iq_format.h"
#include "kernels.h"
#include "trellis_bits.h"

#include <cstdlib>
#include <vector>

namespace atsc_k {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kTile = 16;                          // outputs per tile = one MFMA K-step
constexpr int kTilesPerField = kFieldSyms / kTile; // 16276
constexpr int kKSteps = 8;                         // 128 >= 16 + 110 inputs feed a tile
constexpr int kTapCopy = 144;                      // floats per shifted copy of the padded tap array
constexpr int kTapLead = 16;                       // zeros ahead of tap 0 (p <= 15)
constexpr float kWScale = 65536.0f;                // weights are scaled into fp16's normal range; exact to undo
constexpr int kStageRow = 36;                      // floats per channel row of the output image (32 + 4: conflict-free b128)
constexpr int kMfmaWaves = 4;
constexpr int kSymRow = 136;                       // bytes per channel row of the symbol window (128 + 8: conflict-free b64 reads)

static_assert(kFieldSyms % kTile == 0, "tiles must cover a field exactly");
static_assert(kTile * (kKSteps - 1) + kTile >= kTile + 110, "K must cover the filter");

__device__ float g_taps_mfma[4 * kTapCopy];

// 16 bytes of an output line, non-temporal: the IQ is written once and never read back by the GPU, and kept out of the
// L2's way it leaves the cache to the interleaved bytes and the oscillator table (measured in the fused kernel:
// 2.245 -> 2.181 ms per launch; in a pure store stream the hint changes nothing).
__device__ __forceinline__ void store_line_piece(void* d, const uint4& v) {
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store((v4u){v.x, v.y, v.z, v.w}, reinterpret_cast<v4u*>(d));
}

__device__ __forceinline__ void lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 8 level indices (bytes) -> 8 fp16 levels 2v - 5.75 (signal.h:13-18).  0x6400 | v is the fp16 number 1024 + v.
__device__ __forceinline__ h8 levels8(uint2 w) {
    const h2 k1024 = {(_Float16)1024.0f, (_Float16)1024.0f}, k2 = {(_Float16)2.0f, (_Float16)2.0f},
             kp = {(_Float16)-5.75f, (_Float16)-5.75f};
    h2 r[4];
    const uint32_t src[2] = {w.x, w.y};
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const uint32_t t0 = __builtin_amdgcn_perm(0u, src[d], 0x0c010c00u) | 0x64006400u;
        const uint32_t t1 = __builtin_amdgcn_perm(0u, src[d], 0x0c030c02u) | 0x64006400u;
        r[2 * d] = (__builtin_bit_cast(h2, t0) - k1024) * k2 + kp;
        r[2 * d + 1] = (__builtin_bit_cast(h2, t1) - k1024) * k2 + kp;
    }
    return (h8){r[0].x, r[0].y, r[1].x, r[1].y, r[2].x, r[2].y, r[3].x, r[3].y};
}

template <int FMT> struct Out2;   // two consecutive samples of a channel, one store
template <> struct Out2<kIqCf32> { typedef float4 type; };
template <> struct Out2<kIqCs16> { typedef uint2 type; };
template <> struct Out2<kIqCs8>  { typedef uint32_t type; };

template <int FMT>
__device__ __forceinline__ typename Out2<FMT>::type pack2(float4 v) {
    if constexpr (FMT == kIqCf32) {
        return v;
    } else if constexpr (FMT == kIqCs16) {
        return make_uint2(to_out<kIqCs16>((v2f_t){v.x, v.y}), to_out<kIqCs16>((v2f_t){v.z, v.w}));
    } else {
        return (uint32_t)to_out<kIqCs8>((v2f_t){v.x, v.y}) | ((uint32_t)to_out<kIqCs8>((v2f_t){v.z, v.w}) << 16);
    }
}

}  // namespace

template <int FMT, int NT>
__global__ __launch_bounds__(64 * kMfmaWaves, 2) void k_mix_fir_mfma(const uint8_t* __restrict__ sym, size_t sym_stride,
                                                                     typename OutSample<FMT>::type* __restrict__ iq,
                                                                     size_t iq_stride, const float* __restrict__ planes,
                                                                     typename OutSample<FMT>::type* __restrict__ trash,
                                                                     int channels, int fields, int run8, int runs_per_field,
                                                                     int ch_groups) {
    typedef typename OutSample<FMT>::type out_t;
    typedef typename Out2<FMT>::type out2_t;
    __shared__ __align__(16) float s_taps[4 * kTapCopy];
    __shared__ __align__(16) float s_stage[kMfmaWaves][32 * kStageRow];
    __shared__ __align__(16) uint8_t s_sym[kMfmaWaves][32 * NT * kSymRow];   // 128 symbols x 32*NT channels, rows padded to 136 B
    __shared__ __align__(16) float s_osc[kMfmaWaves][2][2][256];   // double buffer x [re | im] of positions i_group .. +255
    const int tid = threadIdx.x;
    for (int i = tid; i < 4 * kTapCopy; i += 64 * kMfmaWaves) s_taps[i] = g_taps_mfma[i];
    __syncthreads();

    // the wave index is made scalar: task, field, tile and every loop bound below are then wave-uniform SGPR values
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // task = (channel group, run, field); neighbouring waves take neighbouring runs of one field (shared oscillator lines)
    const long task = (long)blockIdx.x * kMfmaWaves + wv;
    const int run = (int)(task % runs_per_field);
    const long g = task / runs_per_field;
    const int cg = (int)(g % ch_groups);
    const int f = (int)(g / ch_groups);
    if (f >= fields) return;
    const int tile0 = run * run8 * 8;

    // A role: row r = 2p + comp
    const int p = r >> 1, comp = r & 1;
    const int sh = (4 - (p & 3)) & 3;                               // which shifted tap copy is 16-byte aligned for this lane
    const float* tapp = s_taps + sh * kTapCopy + (kTapLead + 8 * h - p - sh);   // + 16*ks, + j
    const float* ow0 = s_osc[wv][0][comp] + 8 * h;                              // + 512*(group&1) + 16*u + 16*ks, + j
    // refill of the oscillator window: lane l brings floats 4l..4l+3 of the re plane and of the im plane
    const float* pre_src = planes + (size_t)f * 2 * kPaddedSyms + 4 * lane;
    float* pre_dst = s_osc[wv][0][0] + 4 * lane;                                // + 512*(group&1); re at +0, im at +256
    auto osc_fetch = [&](int tile_first, float4& a, float4& b) {
        // a group that starts past the field is never used: keep its address inside the table (a valid group reads at
        // most up to position 16 * 16272 + 255, inside the padded table)
        const int i = (tile_first < kTilesPerField ? tile_first : 0) * kTile;
        a = *reinterpret_cast<const float4*>(pre_src + i);
        b = *reinterpret_cast<const float4*>(pre_src + kPaddedSyms + i);
    };
    float* stage = s_stage[wv];
    float* st_w = stage + r * kStageRow + 4 * h;                    // + 8*g: outputs 4g+2h, 4g+2h+1 of channel r
    const float* st_r = stage + (lane >> 3) * kStageRow + 4 * (lane & 7);   // + 8*q rows: channel 8q + lane/8, outputs 2(lane&7)..+1

    h8 bw[kKSteps][NT];
    // Symbol window of a group of 8 tiles starting at tile T: positions [16T + 128, 16T + 256) of every channel, i.e. the
    // newest K-step of tiles T+1 .. T+8.  Tile T itself takes the last 16 bytes of the previous window (kept in registers).
    uint8_t* symw = s_sym[wv];
    const uint8_t* sym_rd = symw + r * kSymRow + 8 * h;                         // + 32*nt rows, + 16*(u-1)
    uint8_t* sym_wr = symw + (lane >> 3) * kSymRow + 16 * (lane & 7);           // + 8*q rows
#define ATSC_Q8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
    // lane's 16-byte piece of channel 8q + lane/8 (channels past the end read the last channel; never stored)
#define ATSC_SYM_SRC1(q)                                                                                                 \
    const uint8_t* sym_src##q = sym + (size_t)min(cg * 32 * NT + 8 * q + (lane >> 3), channels - 1) * sym_stride +       \
                                (size_t)f * kFieldSyms + 16 * (lane & 7);
    ATSC_Q8(ATSC_SYM_SRC1)
    // The prefetch registers are named scalars handled by macros on purpose: as arrays indexed in (unrolled) loops they
    // are left in scratch memory by the compiler (a 144-byte private segment, 2 GB of scratch traffic per launch).
#define ATSC_SYM_LOAD1(q) if constexpr (q < 4 * NT) ps##q = *reinterpret_cast<const uint4*>(sym_src##q + src_);
#define ATSC_SYM_FETCH(tile_first)                                                                                      \
    do {                                                                                                                \
        const int tf_ = (tile_first);                                                                                   \
        const int pos_ = (tf_ < kTilesPerField ? tf_ : 0) * kTile + 128 + 16 * (lane & 7);                              \
        const int src_ = (pos_ < kFieldSyms ? pos_ : pos_ - kFieldSyms) - 16 * (lane & 7); /* sync-head copy past the end */ \
        ATSC_Q8(ATSC_SYM_LOAD1)                                                                                         \
    } while (0)
#define ATSC_SYM_PARK1(q) if constexpr (q < 4 * NT) *reinterpret_cast<uint4*>(sym_wr + 8 * q * kSymRow) = ps##q;
#define ATSC_SYM_PARK() do { ATSC_Q8(ATSC_SYM_PARK1) } while (0)
    uint4 ps0, ps1, ps2, ps3, ps4, ps5, ps6, ps7;
    uint2 tl0, tl1;
    ps0 = ps1 = ps2 = ps3 = ps4 = ps5 = ps6 = ps7 = make_uint4(0, 0, 0, 0);
    tl0 = tl1 = make_uint2(0, 0);
    // level window of the run's first tile: the 128 symbols ahead of it, through the same coalesced path
    ATSC_SYM_FETCH(tile0 - 8);
    ATSC_SYM_PARK();
    lds_order();
    // positions [16 T0, 16 T0 + 128): K-steps 0..6 of the first tile, and its newest K-step as the "tail"
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int ks = 0; ks < kKSteps - 1; ++ks)
            bw[ks][nt] = levels8(*reinterpret_cast<const uint2*>(sym_rd + 32 * nt * kSymRow + 16 * ks));
        (nt == 0 ? tl0 : tl1) = *reinterpret_cast<const uint2*>(sym_rd + 32 * nt * kSymRow + 112);
    }
    lds_order();
    ATSC_SYM_FETCH(tile0);
    ATSC_SYM_PARK();
    float4 pre_re, pre_im;
    osc_fetch(tile0, pre_re, pre_im);
    *reinterpret_cast<float4*>(pre_dst) = pre_re;
    *reinterpret_cast<float4*>(pre_dst + 256) = pre_im;
    lds_order();
    for (int t8 = 0; t8 < run8; ++t8) {
        const float* ow = ow0 + 512 * (t8 & 1);
        // next group's oscillator window: loads issued now, parked in the other LDS buffer at the group's last tile
        osc_fetch(tile0 + 8 * (t8 + 1), pre_re, pre_im);
        ATSC_SYM_FETCH(tile0 + 8 * (t8 + 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // The last run of a field is short.  Tiles past the end are computed like any other (their loads stay inside
            // the sync-head copy and the padded oscillator table) and stored to the trash area, as are channels past
            // the last one: the number of loads and stores per tile is fixed, so every vmcnt wait the compiler places
            // is an exact count and never waits for the IQ stores of the previous tile.
            const int tile = tile0 + 8 * t8 + u;
            const bool tile_ok = tile < kTilesPerField;
            const int i0 = tile * kTile;
            // newest K-step of the level window
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const uint2 w = u == 0 ? (nt == 0 ? tl0 : tl1)
                                       : *reinterpret_cast<const uint2*>(sym_rd + 32 * nt * kSymRow + 16 * (u - 1));
                bw[(u + kKSteps - 1) & 7][nt] = levels8(w);
            }
            if (u == 7) {
                // the group's last read of the windows is behind us: keep the tail, then park the next group's windows
                lds_order();
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    (nt == 0 ? tl0 : tl1) = *reinterpret_cast<const uint2*>(sym_rd + 32 * nt * kSymRow + 112);
                lds_order();
                ATSC_SYM_PARK();
                float* d = pre_dst + 512 * ((t8 + 1) & 1);
                *reinterpret_cast<float4*>(d) = pre_re;
                *reinterpret_cast<float4*>(d + 256) = pre_im;
            }
            f16v acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
            // Software pipeline over the K-steps: the weights of step ks+1 are generated (LDS read, multiply, split)
            // while the four MFMAs of step ks run.
            auto weights = [&](int ks, h8& ahi, h8& alo) {
                const float4 o0 = *reinterpret_cast<const float4*>(ow + 16 * u + 16 * ks);
                const float4 o1 = *reinterpret_cast<const float4*>(ow + 16 * u + 16 * ks + 4);
                const float4 c0 = *reinterpret_cast<const float4*>(tapp + 16 * ks);
                const float4 c1 = *reinterpret_cast<const float4*>(tapp + 16 * ks + 4);
                // packed products (v_pk_mul_f32: two per instruction, same rounding as the scalar form)
                const v2f_t xa = (v2f_t){c0.x, c0.y} * (v2f_t){o0.x, o0.y}, xb = (v2f_t){c0.z, c0.w} * (v2f_t){o0.z, o0.w};
                const v2f_t xc = (v2f_t){c1.x, c1.y} * (v2f_t){o1.x, o1.y}, xd = (v2f_t){c1.z, c1.w} * (v2f_t){o1.z, o1.w};
                const float x[8] = {xa.x, xa.y, xb.x, xb.y, xc.x, xc.y, xd.x, xd.y};
                const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                const float oo[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const h2 hh = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x[j], x[j + 1]));
                    // residual of the exact product (one fused multiply-add with the fp16 operand converted in the instruction)
                    const h2 ll = __builtin_bit_cast(
                        h2, __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf(cc[j], oo[j], -(float)hh.x),
                                                       __builtin_fmaf(cc[j + 1], oo[j + 1], -(float)hh.y)));
                    ahi[j] = hh.x; ahi[j + 1] = hh.y;
                    alo[j] = ll.x; alo[j + 1] = ll.y;
                }
            };
            h8 ahi[2], alo[2];
            weights(0, ahi[0], alo[0]);
#pragma unroll
            for (int ks = 0; ks < kKSteps; ++ks) {
                const int cur = ks & 1;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[cur], bw[(u + ks) & 7][nt], acc[nt], 0, 0, 0);
                if (ks + 1 < kKSteps) weights(ks + 1, ahi[cur ^ 1], alo[cur ^ 1]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[cur], bw[(u + ks) & 7][nt], acc[nt], 0, 0, 0);
            }
            // C: register 4g+e of lane (r, h) is row 8g + 4h + e = output 4g + 2h + e/2, component e&1, of channel r
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                constexpr float inv = 1.0f / kWScale;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<float4*>(st_w + 8 * gq) =
                        make_float4(acc[nt][4 * gq] * inv, acc[nt][4 * gq + 1] * inv, acc[nt][4 * gq + 2] * inv,
                                    acc[nt][4 * gq + 3] * inv);
                lds_order();
                const int cbase = cg * 32 * NT + 32 * nt + (lane >> 3);
                out_t* dst = iq + (size_t)f * kFieldSyms + i0 + 2 * (lane & 7);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(st_r + 8 * q * kStageRow);
                    const int c = cbase + 8 * q;
                    out_t* d = (tile_ok && c < channels) ? dst + (size_t)c * iq_stride : trash + 2 * lane;
                    *reinterpret_cast<out2_t*>(d) = pack2<FMT>(v);
                }
                lds_order();
            }
        }
        lds_order();   // the other window buffer is complete and ordered before the next group reads it
    }
}

hipError_t upload_mfma_tables(const float* taps112) {
    std::vector<float> pad(kTapCopy + 4, 0.f), h(4 * kTapCopy);
    for (int k = 0; k < 111; ++k) pad[kTapLead + k] = taps112[k] * kWScale;
    for (int s = 0; s < 4; ++s)
        for (int i = 0; i < kTapCopy; ++i) h[s * kTapCopy + i] = pad[i + s];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_taps_mfma), h.data(), sizeof(float) * h.size());
}

template <int FMT, int NT>
static void launch_nt(hipStream_t s, const uint8_t* d_sym, size_t sym_stride, void* d_iq, size_t iq_stride, const float* planes,
                      void* trash, int channels, int fields, int run8) {
    typedef typename OutSample<FMT>::type out_t;
    const int runs = (kTilesPerField + 8 * run8 - 1) / (8 * run8);
    const int ch_groups = (channels + 32 * NT - 1) / (32 * NT);
    const long tasks = (long)runs * ch_groups * fields;
    const unsigned grid = (unsigned)((tasks + kMfmaWaves - 1) / kMfmaWaves);
    hipLaunchKernelGGL((k_mix_fir_mfma<FMT, NT>), dim3(grid), dim3(64 * kMfmaWaves), 0, s, d_sym, sym_stride,
                       static_cast<out_t*>(d_iq), iq_stride, planes, static_cast<out_t*>(trash), channels, fields, run8, runs,
                       ch_groups);
}

void launch_mix_fir_mfma(hipStream_t s, const uint8_t* d_sym, size_t sym_stride, void* d_iq, size_t iq_stride,
                         const float* osc_planes, int channels, int fields, int iq_format) {
    const int run8 = 4;   // 32 tiles = 512 outputs per wave task
    // One 32-channel tile per wave (NT = 1: 158 VGPRs, 3 waves per SIMD).  Two tiles per wave would share the weight
    // generation but need ~320 VGPRs with the prefetch registers: measured 3.1 ms against 2.4 ms.
    // kOscTabSlack complex64 (8 KB) behind the table: 64 lanes x 16 B of don't-care stores land there
    void* trash = const_cast<float*>(osc_planes) + (size_t)fields * 2 * kPaddedSyms;
#define ATSC_MFMA_CASE(F) launch_nt<F, 1>(s, d_sym, sym_stride, d_iq, iq_stride, osc_planes, trash, channels, fields, run8)
    switch (iq_format) {
        case kIqCs16: ATSC_MFMA_CASE(kIqCs16); break;
        case kIqCs8: ATSC_MFMA_CASE(kIqCs8); break;
        default: ATSC_MFMA_CASE(kIqCf32); break;
    }
#undef ATSC_MFMA_CASE
}


// ======================================================================================= fused trellis/map + FIR
// k_mix_fir_fused: the matrix-core FIR above with the symbol source replaced.  Instead of reading the u8 symbols that
// k_trellis_map wrote to HBM (1 B/symbol written, 1 B/symbol read back: the read stream beside the IQ write stream is
// what holds both FIR forms at 0.87 of the memory system's mixed ceiling), every wave regenerates the symbols of its
// 32 channels from the interleaved bytes (0.25 B/symbol) into a wave-private LDS ring of 256 positions per channel:
// precoder / trellis / mapper / segment sync / field sync exactly as k_trellis_map (reference: trellis.h:10-51,
// field_sync.h:12-26), one byte-time (12 interleaved bytes = 48 consecutive data symbols) per half-wave lane and
// step.  The trellis state entering the run's first byte-time is the part's carry from k_trellis_scan XOR the
// parities of the byte-times of the part before it (recomputed here: <= 206 byte-times, ~6 % of a run's work).
// The symbol tap (d_sym) is optional: with it the ring is also copied out, one 128-byte line per channel and group.
constexpr int kRingRow = 264;                      // bytes per channel row: 256-position ring + 8 (conflict-free b64 reads)
constexpr int kByteTimes = kFieldCoded / 12;       // 5382 byte-times per field, 207 per part

// field-relative position of the first data symbol of byte-time g (uniform)
__device__ __forceinline__ int pos_of_bytetime(int g) {
    if (g >= kByteTimes) return kFieldSyms;
    const int p = g / kPartGroups, tid = g - p * kPartGroups;
    const int t = 4 * tid, sl = t / 69, q = t - 69 * sl;
    return kSegSyms * (1 + 12 * p + sl) + 4 + 12 * q;
}

template <int FMT>
__global__ __launch_bounds__(64 * kMfmaWaves, 2) void k_mix_fir_fused(
    const uint8_t* __restrict__ il, size_t il_stride, const uint4* __restrict__ cin, const uint8_t* __restrict__ last_group,
    const uint8_t* __restrict__ fsync, unsigned parity0, uint8_t* __restrict__ sym_out, size_t sym_stride,
    typename OutSample<FMT>::type* __restrict__ iq, size_t iq_stride, const float* __restrict__ planes,
    typename OutSample<FMT>::type* __restrict__ trash, int channels, int fields, int run8, int runs_per_field,
    int ch_groups) {
    typedef typename OutSample<FMT>::type out_t;
    constexpr int kPieceSamples = 16 / (int)sizeof(out_t);      // samples in a lane's 16-byte piece of a line
    constexpr int kLineTiles = kPieceSamples / 2;               // tiles of 16 samples in a 128-byte line: 1 / 2 / 4
    __shared__ __align__(16) float s_taps[4 * kTapCopy];
    // 62.7 KB per block: two blocks = 8 waves per CU (2 per SIMD, which is also what ~210 VGPRs allow)
    __shared__ __align__(16) float s_stage[kMfmaWaves][32 * kStageRow];    // one 128-byte output line per channel (+ pad)
    __shared__ __align__(16) uint8_t s_ring[kMfmaWaves][32 * kRingRow];
    __shared__ __align__(16) float s_osc[kMfmaWaves][2][256];              // [re | im] of positions i_group .. +255
    const int tid_ = threadIdx.x;
    for (int i = tid_; i < 4 * kTapCopy; i += 64 * kMfmaWaves) s_taps[i] = g_taps_mfma[i];
    __syncthreads();

    // the wave index is made scalar: task, field, tile, byte-time counters and every loop bound below are wave-uniform
    const int lane = tid_ & 63, wv = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int r = lane & 31, h = lane >> 5;
    const long task = (long)blockIdx.x * kMfmaWaves + wv;
    const int run = (int)(task % runs_per_field);
    const long gidx = task / runs_per_field;
    const int cg = (int)(gidx % ch_groups);
    const int f = (int)(gidx / ch_groups);
    if (f >= fields) return;
    const int tile0 = run * run8 * 8;
    const int P0 = tile0 * kTile;

    // ---- FIR roles (as k_mix_fir_mfma)
    const int p = r >> 1, comp = r & 1;
    const int sh = (4 - (p & 3)) & 3;
    // The 8 x 8 taps of this lane do not depend on the tile: registers for the whole run.  (Read from LDS per K-step they
    // are 16 of the ~50 LDS instructions of a tile, the only ones with bank conflicts, and the LDS pipe -- 70-90 % busy --
    // is what bounds this kernel's arithmetic phase.)
    float4 tapr[kKSteps][2];
    {
        const float* tapp = s_taps + sh * kTapCopy + (kTapLead + 8 * h - p - sh);
#pragma unroll
        for (int ks = 0; ks < kKSteps; ++ks) {
            tapr[ks][0] = *reinterpret_cast<const float4*>(tapp + 16 * ks);
            tapr[ks][1] = *reinterpret_cast<const float4*>(tapp + 16 * ks + 4);
        }
    }
    const float* ow = s_osc[wv][comp] + 8 * h;                                  // + 16*u + 16*ks, + j
    const float* pre_src = planes + (size_t)f * 2 * kPaddedSyms + 4 * lane;
    float* pre_dst = s_osc[wv][0] + 4 * lane;                                   // re at +0, im at +256
    auto osc_fetch = [&](int tile_first, float4& a, float4& b) {
        const int i = (tile_first < kTilesPerField ? tile_first : 0) * kTile;
        a = *reinterpret_cast<const float4*>(pre_src + i);
        b = *reinterpret_cast<const float4*>(pre_src + kPaddedSyms + i);
    };
    float* stage = s_stage[wv];
    float* st_w0 = stage + r * kStageRow;                                       // this lane's channel row of the line image
    float* st_w = st_w0 + 4 * h;                                                // complex64: + 8*g: outputs 4g+2h, 4g+2h+1
    const float* st_r = stage + (lane >> 3) * kStageRow + 4 * (lane & 7);       // + 8*q rows

    // ---- symbol producer: lane = (channel r, half h)
    const int ch = min(cg * 32 + r, channels - 1);          // channels past the end recompute the last one; never stored
    const uint8_t* il_c = il + (size_t)ch * il_stride + (size_t)f * kFieldCoded;
    uint8_t* ring = s_ring[wv] + r * kRingRow;
    const uint4* cin_c = cin + ((size_t)ch * fields + f) * kParts;
    const uint32_t* fs_tab = reinterpret_cast<const uint32_t*>(fsync + (((parity0 + (unsigned)f) & 1u) ? kSegSyms : 0));

    // where the run starts (uniform)
    int next_g;        // next byte-time of the field to produce
    int fs_j;          // next field-sync byte to produce, or -1 when not inside the field-sync segment
    int fs_base = 0;   // 0: the field's own sync segment; kFieldSyms: its copy behind the field (field_sync.h:16)
    int prod_pos;      // every (extended) position below it is in the ring
    int part0;
    {
        const int seg = P0 / kSegSyms, jj = P0 - seg * kSegSyms;
        if (seg == 0) {
            next_g = 0; fs_j = jj; prod_pos = P0; part0 = 0;
        } else {
            const int s = seg - 1, pp = s / 12, sl = s - 12 * pp;
            const int j = jj > 4 ? jj - 4 : 0;
            const int t0 = 69 * sl + j / 12;
            next_g = kPartGroups * pp + (t0 >> 2);
            fs_j = -1; part0 = pp;
            prod_pos = P0;   // the byte-time that holds P0 is produced first; earlier symbols of it fall on dead ring slots
        }
    }
    // trellis state entering next_g: the part's carry and the parities of the part's earlier byte-times
    uint32_t cw[3];
    {
        const uint4 ci = cin_c[part0];
        uint32_t acc[3] = {0, 0, 0};
        const int g_lo = part0 * kPartGroups;
        for (int gq = g_lo + h; gq < next_g; gq += 16) {    // eight byte-times per half in flight: the loop is latency-bound
            uint32_t d[8][3];
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) {
                const int gg = min(gq + 2 * k8, kByteTimes - 1);
                const uint32_t* src = reinterpret_cast<const uint32_t*>(il_c) + 3 * gg;
                d[k8][0] = src[0]; d[k8][1] = src[1]; d[k8][2] = src[2];
            }
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) {
                const int gg = gq + 2 * k8;
                if (gg < next_g) {
                    const Words3 e = canonical_group(d[k8], gg);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        uint32_t z2, z0, pr;
                        trellis_local(e.w[k], z2, z0, pr);
                        acc[k] ^= pr;
                    }
                }
            }
        }
        cw[0] = ci.x ^ acc[0] ^ __shfl_xor(acc[0], 32, 64);
        cw[1] = ci.y ^ acc[1] ^ __shfl_xor(acc[1], 32, 64);
        cw[2] = ci.z ^ acc[2] ^ __shfl_xor(acc[2], 32, 64);
    }
    // reserved symbols of the field-sync segment (field_sync.h:17,23), needed by the runs that touch segment 0 or its copy
    uint32_t rsv[3] = {0, 0, 0};
    {
        const uint4 c0 = cin_c[0];
        const uint32_t cinw[3] = {c0.x, c0.y, c0.z};
        const uint32_t* lg = (f == 0) ? reinterpret_cast<const uint32_t*>(last_group + (size_t)ch * 16)
                                      : reinterpret_cast<const uint32_t*>(il_c - 12);
        const Words3 pe = canonical_group(lg, kByteTimes - 1);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            rsv[(k + 1) % 3] = ((cinw[k] & 0x01010101u) << 2) | ((pe.w[k] & 0x01010101u) << 1) | ((cinw[k] >> 2) & 0x01010101u);
    }

    // two byte-times: half h takes next_g + h, whose three interleaved dwords are d0..d2
    auto produce_pair = [&](uint32_t d0, uint32_t d1, uint32_t d2) {
        const int g = next_g + h;
        const bool ok = g < kByteTimes;
        const int gg = ok ? g : kByteTimes - 1;
        const uint32_t d3[3] = {d0, d1, d2};
        const Words3 e = canonical_group(d3, gg);
        uint32_t z2[3], z0[3], pr[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            trellis_local(e.w[k], z2[k], z0[k], pr[k]);
            if (!ok) pr[k] = 0;
        }
        const int pp = gg / kPartGroups, tq = gg - pp * kPartGroups;
        uint32_t hi[3], zz0[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t other = __shfl_xor(pr[k], 32, 64);
            const uint32_t cwk = cw[k] ^ (h ? other : 0u);     // half 1 follows half 0's byte-time
            cw[k] ^= pr[k] ^ other;
            const uint32_t zz2 = z2[k] ^ ((cwk & 0x01010101u) * 0xAAu);
            zz0[k] = z0[k] ^ (((cwk >> 1) & 0x01010101u) * 0x44u) ^ (((cwk >> 2) & 0x01010101u) * 0x11u);
            hi[k] = zz2 | (e.w[k] & 0x55555555u);
        }
        if (ok) {
            // the four symbol times of the byte: segment, place in the segment and commutator phase once per symbol
            // time, then the three 4-encoder words go to slots (k - phase) mod 3
            const int t0 = 4 * tq;
            int sl = t0 / 69, q = t0 - 69 * sl;
            int m = sl % 3;
#pragma unroll
            for (int tau = 0; tau < 4; ++tau) {
                const int seg = 1 + 12 * pp + sl;
                const int base = kSegSyms * seg + 4 + 12 * q;
                if (q == 0) *reinterpret_cast<uint32_t*>(ring + ((kSegSyms * seg) & 255)) = 0x06010106u;
                const int shb = 6 - 2 * tau;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint32_t v = (((hi[k] >> shb) & 0x03030303u) << 1) | ((zz0[k] >> shb) & 0x01010101u);
                    int hslot = k - m;
                    hslot += hslot < 0 ? 3 : 0;
                    *reinterpret_cast<uint32_t*>(ring + ((base + 4 * hslot) & 255)) = v;
                }
                if (++q == 69) { q = 0; ++sl; m = m == 2 ? 0 : m + 1; }
            }
        }
        next_g += 2;
        if (next_g >= kByteTimes) {
            fs_j = 0; fs_base = kFieldSyms; prod_pos = kFieldSyms;    // behind the field: the copy of its sync head
        } else {
            prod_pos = pos_of_bytetime(next_g) - 4;
        }
    };

    // everything below `limit` (extended position) into the ring
    // (the first `npre` pairs of byte-times take their interleaved dwords from the prefetch registers)
    auto produce_until = [&](int limit, int npre, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t b0, uint32_t b1,
                             uint32_t b2) {
        int it = 0;
        while (prod_pos < limit) {
            if (fs_j >= 0) {
                // field-sync segment: the table, its last three dwords replaced by the reserved symbols; both halves
                // of a channel's lane pair write alternate dwords
                const int n = min(kSegSyms - fs_j, ((limit - prod_pos) + 3) & ~3);
                for (int d = (fs_j >> 2) + h; d < ((fs_j + n) >> 2); d += 2) {
                    const uint32_t v = d < 205 ? fs_tab[d] : (d == 205 ? rsv[0] : (d == 206 ? rsv[1] : rsv[2]));
                    *reinterpret_cast<uint32_t*>(ring + ((fs_base + 4 * d) & 255)) = v;
                }
                fs_j += n;
                prod_pos += n;
                if (fs_j >= kSegSyms) {
                    fs_j = -1;
                    if (fs_base) prod_pos = 1 << 30;     // the copy behind the field is complete: nothing further exists
                    else prod_pos = kSegSyms;             // data segment 1 follows (its sync comes with byte-time 0)
                }
                continue;
            }
            uint32_t d0, d1, d2;
            if (it < npre) {
                d0 = it ? b0 : a0; d1 = it ? b1 : a1; d2 = it ? b2 : a2;
            } else {
                const uint32_t* src = reinterpret_cast<const uint32_t*>(il_c) + 3 * min(next_g + h, kByteTimes - 1);
                d0 = src[0]; d1 = src[1]; d2 = src[2];
            }
            ++it;
            produce_pair(d0, d1, d2);
        }
    };

    // symbol tap: positions [pos, pos + 128) of the ring -> d_sym, one 128-byte line per channel (8 lanes x 16 B)
    // (every lane always stores, like the IQ path: pieces that do not exist go to the trash area)
    auto tap = [&](int pos, int pieces) {
        if (!sym_out) return;
        const int pp = pos + 16 * (lane & 7);
        const bool live = (lane & 7) < pieces && pp < kFieldSyms;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = cg * 32 + 8 * q + (lane >> 3);
            const uint8_t* row = s_ring[wv] + (8 * q + (lane >> 3)) * kRingRow + (pp & 255);
            const uint2 a = *reinterpret_cast<const uint2*>(row), b = *reinterpret_cast<const uint2*>(row + 8);
            uint8_t* d = (live && c < channels) ? sym_out + (size_t)c * sym_stride + (size_t)f * kFieldSyms + pp
                                                : reinterpret_cast<uint8_t*>(trash) + 16 * lane;
            store_line_piece(d, make_uint4(a.x, a.y, b.x, b.y));
        }
    };

    // ---- prime: symbols [P0, P0 + 112) for K-steps 0..6 of the first tile
    produce_until(P0 + 112, 0, 0, 0, 0, 0, 0, 0);
    lds_order();
    h8 bw[kKSteps][1];
#pragma unroll
    for (int ks = 0; ks < kKSteps - 1; ++ks)
        bw[ks][0] = levels8(*reinterpret_cast<const uint2*>(ring + ((P0 + 16 * ks + 8 * h) & 255)));
    lds_order();
    produce_until(P0 + 240 + 16, 0, 0, 0, 0, 0, 0, 0);
    lds_order();
    if (run == 0) tap(0, 8);         // [0, 128): the field's first symbols, which no group's tap covers

    float4 pre_re, pre_im;
    osc_fetch(tile0, pre_re, pre_im);
    *reinterpret_cast<float4*>(pre_dst) = pre_re;
    *reinterpret_cast<float4*>(pre_dst + 256) = pre_im;
    lds_order();
    for (int t8 = 0; t8 < run8; ++t8) {
        const int Tg = tile0 + 8 * t8;
        // The interleaved bytes of the next four byte-times are fetched NOW and used at the group's last tile.  vmcnt
        // completes in order: a load issued at production time would make the wave wait for every IQ store of the group
        // before it (that drain cost 0.6 ms per launch); a load issued here only waits for the previous group's stores,
        // seven tile times old.  The two pairs are consumed by a fixed two-trip loop without loads so that the
        // compiler's wait is an exact count.
#ifndef ATSC_FUSED_NOLOADS
        osc_fetch(Tg + 8, pre_re, pre_im);             // next group's oscillator window, parked after this group's last tile
#endif
        const uint32_t* ia = reinterpret_cast<const uint32_t*>(il_c) + 3 * min(next_g + h, kByteTimes - 1);
        const uint32_t* ib = reinterpret_cast<const uint32_t*>(il_c) + 3 * min(next_g + 2 + h, kByteTimes - 1);
#ifdef ATSC_FUSED_NOLOADS   // timing experiment: no global loads in the steady state
        const uint32_t a0 = (uint32_t)(size_t)ia, a1 = a0 * 3, a2 = a0 * 5, b0 = (uint32_t)(size_t)ib, b1 = b0 * 3, b2 = b0 * 7;
#else
        const uint32_t a0 = ia[0], a1 = ia[1], a2 = ia[2], b0 = ib[0], b1 = ib[1], b2 = ib[2];
#endif
        const bool pre_ok = fs_j < 0;
        __builtin_amdgcn_sched_barrier(0);
        // [16Tg + 128, 16Tg + 256): produced, not yet overwritten, and whole 128-byte lines of the symbol rows (a tap that
        // straddles lines writes every line in two halves, 9 us apart: 0.8 ms per launch)
        if (Tg < kTilesPerField) tap(16 * Tg + 128, 8);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int tile = Tg + u;
            const bool tile_ok = tile < kTilesPerField;
            (void)tile_ok;
            const int i0 = tile * kTile;
            bw[(u + kKSteps - 1) & 7][0] = levels8(*reinterpret_cast<const uint2*>(ring + ((i0 + 112 + 8 * h) & 255)));
            if (u == 7) {
                // tile 7 has taken its symbols: the ring may advance to what the next group needs
                lds_order();
                const int limit = 16 * (Tg + 8) + 240 + 16;
#ifdef ATSC_FUSED_NOCOMPUTE
                prod_pos = 1 << 30;
#endif
#pragma unroll 1
                for (int it = 0; it < 2; ++it)
                    if (pre_ok && fs_j < 0 && prod_pos < limit) produce_pair(it ? b0 : a0, it ? b1 : a1, it ? b2 : a2);
                produce_until(limit, 0, 0, 0, 0, 0, 0, 0);   // field sync, field end, a third pair at a segment boundary
            }
            f16v acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            auto weights = [&](int ks, h8& ahi, h8& alo) {
                const float4 o0 = *reinterpret_cast<const float4*>(ow + 16 * u + 16 * ks);
                const float4 o1 = *reinterpret_cast<const float4*>(ow + 16 * u + 16 * ks + 4);
                const float4 c0 = tapr[ks][0], c1 = tapr[ks][1];
                // packed products (v_pk_mul_f32: two per instruction, same rounding as the scalar form)
                const v2f_t xa = (v2f_t){c0.x, c0.y} * (v2f_t){o0.x, o0.y}, xb = (v2f_t){c0.z, c0.w} * (v2f_t){o0.z, o0.w};
                const v2f_t xc = (v2f_t){c1.x, c1.y} * (v2f_t){o1.x, o1.y}, xd = (v2f_t){c1.z, c1.w} * (v2f_t){o1.z, o1.w};
                const float x[8] = {xa.x, xa.y, xb.x, xb.y, xc.x, xc.y, xd.x, xd.y};
                const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                const float oo[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const h2 hh = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x[j], x[j + 1]));
                    // residual of the exact product (one fused multiply-add with the fp16 operand converted in the instruction)
                    const h2 ll = __builtin_bit_cast(
                        h2, __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf(cc[j], oo[j], -(float)hh.x),
                                                       __builtin_fmaf(cc[j + 1], oo[j + 1], -(float)hh.y)));
                    ahi[j] = hh.x; ahi[j + 1] = hh.y;
                    alo[j] = ll.x; alo[j + 1] = ll.y;
                }
            };
            // (one weight buffer: the other wave of the SIMD covers the generation latency)
            // The LDS image holds one 128-byte line per channel in the OUTPUT format: 16 complex64 samples = this tile, 32
            // CS16 samples = two tiles, 64 CS8 samples = four (kLineTiles) -- a fixed-point format stored tile by tile writes
            // every line in halves / quarters, and partial lines are what the memory side is slow at.  One quarter of the
            // PREVIOUS line (8 channels x 128 B = 1 KB) leaves after every second K-step of the tile that follows it: four
            // stores spread over a tile instead of a burst at its end (a burst fills the CU's store queue and the waves
            // behind it stop at their own stores with the matrix pipe idle: 2.30 -> 2.255 ms).
            auto out_store = [&](int q, const uint4& v, int ltile, bool ok) {
                const int c = cg * 32 + 8 * q + (lane >> 3);
                out_t* d = (ok && c < channels)
                               ? iq + (size_t)c * iq_stride + (size_t)f * kFieldSyms + ltile * kTile + kPieceSamples * (lane & 7)
                               : reinterpret_cast<out_t*>(reinterpret_cast<uint4*>(trash) + lane);
#ifdef ATSC_FUSED_TRASH   // timing experiment: compute floor (every store into the cache-resident trash area)
                d = reinterpret_cast<out_t*>(reinterpret_cast<uint4*>(trash) + lane);
#endif
#ifdef ATSC_FUSED_NOSTORE   // timing experiment: no store is executed (the condition is never true, the compiler cannot know)
                if (channels < 0)
#endif
                store_line_piece(d, v);
            };
            const bool store_tile = (u % kLineTiles) == 0;    // this tile's K loop carries the previous line's stores
#ifdef ATSC_FUSED_NOCOMPUTE   // timing experiment: the store stream alone, in this kernel's own structure
            for (int e = 0; e < 4; ++e) acc[e] = (float)bw[(u + e) & 7][0][e];
#else
            const bool prev_ok = (u >= kLineTiles || t8 > 0) && tile - kLineTiles < kTilesPerField;
            uint4 sv;
#ifdef ATSC_FUSED_HALFGEN   // timing experiment (wrong results): every second K-step reuses the previous step's weights
            h8 ahi, alo;
#endif
#pragma unroll
            for (int ks = 0; ks < kKSteps; ++ks) {
#ifdef ATSC_FUSED_HALFGEN
                if ((ks & 1) == 0) weights(ks, ahi, alo);
#else
                h8 ahi, alo;
                weights(ks, ahi, alo);
#endif
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bw[(u + ks) & 7][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bw[(u + ks) & 7][0], acc, 0, 0, 0);
                if (store_tile) {
                    if ((ks & 1) == 0) sv = *reinterpret_cast<const uint4*>(st_r + 8 * (ks >> 1) * kStageRow);
                    else {
                        out_store(ks >> 1, sv, tile - kLineTiles, prev_ok);
                        __builtin_amdgcn_sched_barrier(0);    // the scheduler would otherwise sink the stores to the tile's end
                    }
                }
            }
#endif
            if (u == 7) {
                // the group's last read of the oscillator window is issued: the next group's window replaces it
                lds_order();
                *reinterpret_cast<float4*>(pre_dst) = pre_re;
                *reinterpret_cast<float4*>(pre_dst + 256) = pre_im;
            }
            // C -> this tile's part of the line image (the image's last quarter was read at K-step 6 of a store tile)
            constexpr float inv = 1.0f / kWScale;
            const int ph = u % kLineTiles;
            lds_order();
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                // outputs 4gq + 2h and 4gq + 2h + 1 of channel r: two complex samples
                const v2f_t s0 = {acc[4 * gq] * inv, acc[4 * gq + 1] * inv}, s1 = {acc[4 * gq + 2] * inv, acc[4 * gq + 3] * inv};
                if constexpr (FMT == kIqCf32) {
                    *reinterpret_cast<float4*>(st_w + 8 * gq) = make_float4(s0.x, s0.y, s1.x, s1.y);
                } else if constexpr (FMT == kIqCs16) {
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint32_t*>(st_w0) + 16 * ph + 4 * gq + 2 * h) =
                        make_uint2(to_out<kIqCs16>(s0), to_out<kIqCs16>(s1));
                } else {
                    *(reinterpret_cast<uint32_t*>(st_w0) + 8 * ph + 2 * gq + h) =
                        (uint32_t)to_out<kIqCs8>(s0) | ((uint32_t)to_out<kIqCs8>(s1) << 16);
                }
            }
            lds_order();
        }
        lds_order();
    }
    {   // the run's last line
        const int ltile = tile0 + 8 * run8 - kLineTiles;
        const int cq = cg * 32 + (lane >> 3);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 v = *reinterpret_cast<const uint4*>(st_r + 8 * q * kStageRow);
            const int c = cq + 8 * q;
            out_t* d = (ltile < kTilesPerField && c < channels)
                           ? iq + (size_t)c * iq_stride + (size_t)f * kFieldSyms + ltile * kTile + kPieceSamples * (lane & 7)
                           : reinterpret_cast<out_t*>(reinterpret_cast<uint4*>(trash) + lane);
#ifdef ATSC_FUSED_NOSTORE
            if (channels < 0)
#endif
            store_line_piece(d, v);
        }
    }
}

template <int FMT>
static void launch_fused_fmt(hipStream_t s, const uint8_t* il, size_t il_stride, const uint4* cin, const uint8_t* last_group,
                             const uint8_t* fsync, unsigned parity0, uint8_t* d_sym, size_t sym_stride, void* d_iq,
                             size_t iq_stride, const float* planes, void* trash, int channels, int fields) {
    typedef typename OutSample<FMT>::type out_t;
    static const int run8_env = [] { const char* e = getenv("ATSC_FUSED_RUN8"); return e ? atoi(e) : 0; }();
    const int ch_groups = (channels + 31) / 32;
    // Long runs amortise what a run pays at its start (carry prefix of up to 206 byte-times, priming the ring: about 12
    // tile times); short runs balance the grid.  The chip holds 2048 of these waves at once (2 per SIMD), so a launch takes
    // about ceil(tasks / 2048) rounds of (run length + 12) tile times: the run length that minimises that (measured, 64
    // channels x 64 fields: 2.39 ms at 128 groups per run = one round, 2.43 at 64, 2.48 at 32, 2.61 at 16).
    int run8 = run8_env;
    if (run8 <= 0) {
        long best = -1;
        for (int cand = 128; cand >= 8; cand >>= 1) {
            const long t = (long)((kTilesPerField + 8 * cand - 1) / (8 * cand)) * ch_groups * fields;
            const long cost = ((t + 2047) / 2048) * (8 * cand + 12);
            if (best < 0 || cost < best) { best = cost; run8 = cand; }
        }
    }
    const int runs = (kTilesPerField + 8 * run8 - 1) / (8 * run8);
    const long tasks = (long)runs * ch_groups * fields;
    const unsigned grid = (unsigned)((tasks + kMfmaWaves - 1) / kMfmaWaves);
    hipLaunchKernelGGL((k_mix_fir_fused<FMT>), dim3(grid), dim3(64 * kMfmaWaves), 0, s, il, il_stride, cin, last_group, fsync,
                       parity0, d_sym, sym_stride, static_cast<out_t*>(d_iq), iq_stride, planes, static_cast<out_t*>(trash),
                       channels, fields, run8, runs, ch_groups);
}

void launch_mix_fir_fused(hipStream_t s, const DeviceTables& t, const Buffers& b, uint8_t* d_sym, size_t sym_stride, void* d_iq,
                          size_t iq_stride, const float* osc_planes, int channels, int fields, uint64_t field_base,
                          int iq_format) {
    void* trash = const_cast<float*>(osc_planes) + (size_t)fields * 2 * kPaddedSyms;
    const unsigned par0 = (unsigned)(field_base & 1);
    switch (iq_format) {
        case kIqCs16:
            launch_fused_fmt<kIqCs16>(s, b.il, b.il_stride, b.cin, b.last_group, t.fsync, par0, d_sym, sym_stride, d_iq, iq_stride,
                                      osc_planes, trash, channels, fields);
            break;
        case kIqCs8:
            launch_fused_fmt<kIqCs8>(s, b.il, b.il_stride, b.cin, b.last_group, t.fsync, par0, d_sym, sym_stride, d_iq, iq_stride,
                                     osc_planes, trash, channels, fields);
            break;
        default:
            launch_fused_fmt<kIqCf32>(s, b.il, b.il_stride, b.cin, b.last_group, t.fsync, par0, d_sym, sym_stride, d_iq, iq_stride,
                                      osc_planes, trash, channels, fields);
            break;
    }
}

}  // namespace atsc_k
