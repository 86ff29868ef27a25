// Host-side generators for the constant tables of the 8VSB path.  Built once per handle and
// uploaded to HBM; kernels never recompute them.
#pragma once
#include <cstdint>
#include <vector>

namespace atsc_tables {

constexpr int kPackets = 312;
constexpr int kTsBytes = 188;
constexpr int kSegData = 187;
constexpr int kSegCoded = 207;
constexpr int kParity = 20;
constexpr int kFieldTs = kPackets * kTsBytes;        // 58656
constexpr int kFieldCoded = kPackets * kSegCoded;    // 64584
constexpr int kSegSyms = 832;
constexpr int kFieldSyms = 313 * kSegSyms;           // 260416
constexpr int kPaddedSyms = kFieldSyms + kSegSyms;   // 261248
constexpr int kTail = 208 * 51;                      // 10608
constexpr int kTaps = 112;
constexpr int kRsLdsBytes = 65536 + 32768;

struct Tables {
    // PRBS laid out like the TS field (188 B per packet, byte 0 of each packet = 0 so the sync byte
    // passes through the XOR untouched before it is dropped): lets the kernel XOR whole dwords.
    std::vector<uint8_t> prbs188;   // 58656
    std::vector<uint8_t> prbs;      // 58344, plain order (reference: randomize.h:17-34)
    // RS(207,187) feedback rows: row[fb] = { g[19]*fb, g[18]*fb, ..., g[0]*fb } padded to 32 B.
    std::vector<uint8_t> rs_rows;   // 256*32
    // Four-bytes-per-step form: rs_rows4[m-1][v] = state after byte v followed by m-1 zero bytes (32-byte rows).
    std::vector<uint8_t> rs_rows4;  // 4*256*32
    std::vector<uint8_t> rs_lds;    // kRsLdsBytes: rs_rows4 in the bank-spread LDS layout of k_rs_parity
    std::vector<uint8_t> prbs_parity;  // 312 x 32 B: RS parity of each PRBS row (bytes 0..19)
    uint8_t rs_gen[21];             // g[0..20] ascending
    std::vector<uint8_t> fsync[2];  // 832 B each: field sync segment, even / odd, reserved 12 = 0
    float taps[kTaps];
    float taps_half[kTaps];   // the same pulse half a symbol later: 112 taps centred between taps 55 and 56 (2x output)
    std::vector<float> osc0;        // 261248 * 2
    float scale[2];
};

void build(Tables& t);

}  // namespace atsc_tables
