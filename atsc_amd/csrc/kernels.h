// Launch interface between the C-ABI host layer (atsc_hip.cpp) and the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <vector>

namespace atsc_k {

// Geometry of the path (reference: src/common/atsc_parameters.h:5-25).
constexpr int kPackets = 312;
constexpr int kTsBytes = 188;
constexpr int kSegCoded = 207;
constexpr int kFieldTs = kPackets * kTsBytes;      // 58656
constexpr int kFieldCoded = kPackets * kSegCoded;  // 64584
constexpr int kSegSyms = 832;
constexpr int kFieldSyms = 313 * kSegSyms;         // 260416
constexpr int kPaddedSyms = kFieldSyms + kSegSyms; // 261248
constexpr int kTail = 208 * 51;                    // 10608 coded bytes of history the interleaver needs
constexpr int kParts = 26;                         // trellis "parts" per field: 12 segments = 207 byte-times each
constexpr int kPartGroups = 207;                   // 12-byte groups (byte-times) per part
constexpr int kTaps = 112;

// Overlap-save FIR geometry (fir_fft.hip): 1024-point blocks advancing by 896 outputs.
constexpr int kFftN = 1024;
constexpr int kFftValid = 896;
constexpr int kFftBlocks = (kFieldSyms + kFftValid - 1) / kFftValid;  // 291
constexpr int kIqCf32 = 0, kIqCs16 = 1, kIqCs8 = 2;   // == ATSC_HIP_IQ_* (atsc_hip.h)
constexpr int kOscTabSlack = 1024;  // complex64 entries behind the oscillator table that absorb stores of rows that do not exist

struct DeviceTables {
    const uint32_t* prbs188;   // 58656 B, TS-field layout, dword view
    const uint4* rs_lds;       // 98304 B: RS lookup tables in the LDS image of k_rs_parity (tables.h rs_lds)
    const uint4* prbs_parity;  // 312 x 32 B: RS parity of every PRBS row
    const uint8_t* fsync;      // 2 x 832 B (even, odd)
    const uint8_t* prbs;       // 58344 B, plain order (187 per packet)
    const uint8_t* prbs_rows;  // 312 x 208 B: the PRBS of a packet's 187 data bytes, then zeros over the parity bytes
};

struct Buffers {
    // [channel] strides in bytes
    uint8_t* coded;  size_t coded_stride;   // debug tap only (ATSC_HIP_FLAG_TAPS): per channel [kTail unused | fields*64584]
    uint8_t* tail;                           // [channel][kTail]: last 10608 RS-coded bytes of the previous batch
    uint8_t* tail_next;                      // the same of THIS batch, written by k_interleave while other workgroups still read `tail`;
                                             // the host swaps the two behind launch_interleave
    uint4* parity;                           // [channel][fields*312] x 32 B: RS parity per packet (20 bytes used)
    uint8_t* il;     size_t il_stride;      // per channel: fields*64584, transmission order
    uint4* par;                              // [channel][fields*26] parities (3 words used)
    uint4* cin;                              // [channel][fields*26] carries entering each part
    uint4* carry;                            // [channel] trellis state entering the batch (updated by the scan)
    uint4* span_par;                         // [channel] XOR of all parities of the batch
    uint8_t* last_group;                     // [channel][16] last 12 interleaved bytes of the previous batch (what this batch's first field starts from)
    uint8_t* last_group_next;                // the same for the NEXT batch: written by this batch's save (launch_save_state /
                                             // launch_scan_osc_state) while the FIR may still read last_group; the host swaps the two
    int max_fields;
};

void launch_rand_rs(hipStream_t s, const DeviceTables& t, const uint8_t* d_ts, size_t ts_stride, const Buffers& b,
                    int channels, int fields);
// interleaver + per-part trellis parities straight from TS + parity (+ the saved coded tail b.tail); leaves the batch's own
// tail in b.tail_next: the caller swaps the two pointers afterwards
void launch_interleave(hipStream_t s, const DeviceTables& t, const uint8_t* d_ts, size_t ts_stride, const Buffers& b,
                       int channels, int fields);
// debug tap: materialises the RS-coded stream b.coded (not needed by the path itself)
void launch_assemble(hipStream_t s, const DeviceTables& t, const uint8_t* d_ts, size_t ts_stride, const Buffers& b,
                     int channels, int fields);
void launch_trellis_scan(hipStream_t s, const Buffers& b, const uint4* carry_xor, int channels, int fields);
// the fused path: carry scan + oscillator table of the batch (launch_osc_table's arguments) + save of the last interleaved bytes
// into b.last_group_next, as ONE launch
void launch_scan_osc_state(hipStream_t s, const Buffers& b, const uint4* carry_xor, int channels, int fields, const float2* osc_in,
                           float2* osc_out, float2* osc_tab, float2 scale, int planes);
void launch_span_parity(hipStream_t s, const Buffers& b, uint4* d_out, int channels, int fields);
void launch_trellis_map(hipStream_t s, const DeviceTables& t, const Buffers& b, uint8_t* d_sym, size_t sym_stride,
                        int channels, int fields, uint64_t field_base);
void launch_save_state(hipStream_t s, const Buffers& b, int channels, int fields);
// Oscillator state alone advanced by `steps` fields (offset.h:35-41).  261248 complex64 each.
void launch_osc_advance(hipStream_t s, const float2* osc_in, float2* osc_out, float2 scale, unsigned steps);
// Mixer + FIR (fir_fft.hip).  osc_tab: (fields x 261248 + kOscTabSlack) complex64 scratch.
// launch_osc_table fills osc_tab[f] = T_(f0+f) for the batch from the state osc_in and leaves the state after the
// batch in osc_out (offset.h:35-41); launch_mix_fir_fft then produces the IQ of every (channel, field).
// planes: 0 = complex64 entries (FFT form); kOscPlanes4096 / kOscPlanes16 = the fp16-piece planes of the matrix-core forms, the
// entries scaled by 2^12 (fir_fused.hip) or 2^4 (fir_w.hip)
constexpr int kOscComplex = 0, kOscPlanes4096 = 1, kOscPlanes16 = 2;
void launch_osc_table(hipStream_t s, const float2* osc_in, float2* osc_out, float2* osc_tab, float2 scale, int fields,
                      int planes);
// iq_format: kIqCf32 (complex64, the reference's payload) / kIqCs16 / kIqCs8 (SDR sink formats); iq_stride in samples.
void launch_mix_fir_fft(hipStream_t s, const uint8_t* d_sym, size_t sym_stride, void* d_iq, size_t iq_stride,
                        float2* osc_tab, int channels, int fields, int iq_format);
hipError_t upload_fft_tables(const float* taps112);
// Mixer + FIR on the matrix cores for handles with many channels (fir_fused.hip, the form without the symbol producer): osc_planes is the osc_tab memory
// filled by launch_osc_table(..., planes = true).
// upsample = 2: the interpolating form, 2 x 260416 samples per field (iq_stride counts output samples)
void launch_mix_fir_mfma(hipStream_t s, const uint8_t* d_sym, size_t sym_stride, void* d_iq, size_t iq_stride,
                         const float* osc_planes, int channels, int fields, int iq_format, int upsample);
hipError_t upload_mm_tables(const float* taps112, const float* taps_half112);   // fir_fused.hip: the Toeplitz tap operands
// The operand itself, as fp16 bit patterns in the kernel's register order [phase][hi, lo][K-step 0..8][lane 0..63][8]: lane
// (p = lane & 31, h = lane >> 5) of K-step ks holds taps[16 ks + 8 h + j - p] x 2^16.  build_: on the host (no GPU);
// download_: read back from the current device's copy.
constexpr size_t kMmOperandHalves = 2 * 2 * 9 * 64 * 8;
void build_mm_operand(const float* taps112, const float* taps_half112, std::vector<uint16_t>& out);
hipError_t download_mm_tables(uint16_t* out);
// The same with the trellis / map / sync stage fused in: symbols are regenerated from b.il + b.cin inside the kernel and
// never pass through HBM; d_sym (the symbol tap) may be NULL.  Replaces launch_trellis_map + launch_mix_fir_mfma.
// ev != NULL: the kernel's own dispatch records ev->start / ev->stop (no event packets around it).
struct KernelEvents { hipEvent_t start, stop; };
void launch_mix_fir_fused(hipStream_t s, const DeviceTables& t, const Buffers& b, uint8_t* d_sym, size_t sym_stride, void* d_iq,
                          size_t iq_stride, const float* osc_planes, int channels, int fields, uint64_t field_base,
                          int iq_format, int upsample, const KernelEvents* ev = nullptr);
// The fused kernel with the oscillator folded into the tap operand (fir_w.hip): 64 channels per wave task, complex64 output,
// one sample per symbol.  osc_planes: launch_osc_table(..., kOscPlanes16).
void launch_mix_fir_wide(hipStream_t s, const DeviceTables& t, const Buffers& b, uint8_t* d_sym, size_t sym_stride, void* d_iq,
                         size_t iq_stride, const float* osc_planes, int channels, int fields, uint64_t field_base);
hipError_t upload_w_tables(const float* taps112);
// fir_w.hip's tap operand [hi, lo][K-step 0..7][lane][8]: lane (p = (lane & 31) >> 1, h = lane >> 5) holds taps[16 ks + 8 h + j - p] x 2^8
constexpr size_t kWOperandHalves = 2 * 8 * 64 * 8;
void build_w_operand(const float* taps112, std::vector<uint16_t>& out);
hipError_t download_w_tables(uint16_t* out);
constexpr int kWideGroup = 64;    // channels per wave task of fir_w.hip
constexpr int kFusedGroup = 16;   // channels per wave task of the fused kernel: a partly filled group costs as much as a full one

}  // namespace atsc_k
