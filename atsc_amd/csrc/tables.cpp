// Constant tables of the 8VSB modulator path (host side).  Each generator cites the reference
// behaviour it must reproduce bit-for-bit; tests/test_cabi_cpu.py::test_tables_match_oracle_and_golden
// checks them against the oracle and the golden fixtures.
#include "tables.h"

#include <cmath>
#include <cstring>

namespace atsc_tables {

namespace {

// ---- randomizer PRBS (reference: src/atsc/randomize.h:13-33).  16-bit register seeded 0xf180,
// feedback 0x9c65 applied after the shift; eight fixed register bits form the output byte.
void build_prbs(Tables& t) {
    t.prbs.resize(kSegData * kPackets);
    t.prbs188.assign(kFieldTs, 0);
    uint32_t reg = 0xf180;
    for (int s = 0; s < kPackets; ++s) {
        for (int k = 0; k < kSegData; ++k) {
            const uint8_t out = static_cast<uint8_t>(((reg >> 6) & 0xf0) | ((reg >> 3) & 0x08) | ((reg >> 1) & 0x06) | (reg & 1));
            t.prbs[s * kSegData + k] = out;
            t.prbs188[s * kTsBytes + 1 + k] = out;
            reg <<= 1;
            if (reg & 0x10000) reg ^= 0x138cb;  // (0x9c65 << 1) | 1, also clears bit 16
        }
    }
}

// ---- GF(256)/0x11d and the RS(207,187) generator, roots alpha^0..alpha^19
// (reference: src/atsc/galois.h:85-113, src/atsc/reed_solomon.h:23-39).
struct Gf {
    uint8_t exp[510];
    uint8_t log[256];
    Gf() {
        unsigned v = 1;
        for (int i = 0; i < 255; ++i) {
            exp[i] = exp[i + 255] = static_cast<uint8_t>(v);
            log[v] = static_cast<uint8_t>(i);
            v = (v << 1) ^ ((v & 0x80) ? 0x11d : 0);
        }
        log[0] = 0;
    }
    uint8_t mul(uint8_t a, uint8_t b) const { return (a && b) ? exp[log[a] + log[b]] : 0; }
};

void build_rs(Tables& t) {
    Gf gf;
    uint8_t g[kParity + 1] = {1};
    for (int i = 0; i < kParity; ++i) {
        const uint8_t root = gf.exp[i];
        for (int j = i + 1; j >= 1; --j) g[j] = g[j - 1] ^ gf.mul(g[j], root);
        g[0] = gf.mul(g[0], root);
    }
    std::memcpy(t.rs_gen, g, sizeof g);
    // Feedback rows for the systematic LFSR (reference: reed_solomon.h:192-209): after the shift,
    // parity byte i receives g[19-i]*fb.
    t.rs_rows.assign(256 * 32, 0);
    for (int fb = 0; fb < 256; ++fb)
        for (int i = 0; i < kParity; ++i) t.rs_rows[fb * 32 + i] = gf.mul(g[kParity - 1 - i], static_cast<uint8_t>(fb));
    // Four data bytes per step.  One LFSR step is s' = L(s ^ x e0) with the linear map L(v) = shift(v) ^ row[v[0]], so
    // after four steps   s4 = shift4(s) ^ T4[s0^d0] ^ T3[s1^d1] ^ T2[s2^d2] ^ T1[s3^d3],   Tm[v] = L^m(v e0), T1 = row.
    // Stored [m-1][v] as 32-byte rows.
    t.rs_rows4.assign(4 * 256 * 32, 0);
    for (int v = 0; v < 256; ++v) {
        uint8_t st[kParity] = {static_cast<uint8_t>(v)};
        for (int m = 1; m <= 4; ++m) {
            const uint8_t* row = &t.rs_rows[st[0] * 32];
            uint8_t nx[kParity];
            for (int i = 0; i < kParity; ++i) nx[i] = static_cast<uint8_t>((i + 1 < kParity ? st[i + 1] : 0) ^ row[i]);
            std::memcpy(st, nx, sizeof st);
            std::memcpy(&t.rs_rows4[((m - 1) * 256 + v) * 32], st, kParity);
        }
    }
    // The same four tables in the bank-spread LDS image of k_rs_parity: the first 16 bytes of row (m, v) four times
    // (copy = lane & 3) at 16-byte index (m*256 + v)*4 + copy, then the last 4 bytes eight times (copy = lane & 7) at
    // dword index (m*256 + v)*8 + copy.  Lanes that share a copy spread over (v mod 4) only.
    t.rs_lds.assign(kRsLdsBytes, 0);
    for (int m = 0; m < 4; ++m)
        for (int v = 0; v < 256; ++v) {
            const uint8_t* row = &t.rs_rows4[(m * 256 + v) * 32];
            for (int c = 0; c < 4; ++c) std::memcpy(&t.rs_lds[((m * 256 + v) * 4 + c) * 16], row, 16);
            for (int c = 0; c < 8; ++c) std::memcpy(&t.rs_lds[65536 + ((m * 256 + v) * 8 + c) * 4], row + 16, 4);
        }
    // RS is linear, so the kernel encodes the raw TS bytes and adds the parity of the PRBS row afterwards.
    t.prbs_parity.assign(kPackets * 32, 0);
    for (int s = 0; s < kPackets; ++s) {
        uint8_t st[kParity] = {0};
        for (int k = 0; k < kSegData; ++k) {
            const uint8_t fb = static_cast<uint8_t>(t.prbs[s * kSegData + k] ^ st[0]);
            const uint8_t* row = &t.rs_rows[fb * 32];
            for (int i = 0; i < kParity; ++i) st[i] = static_cast<uint8_t>((i + 1 < kParity ? st[i + 1] : 0) ^ row[i]);
        }
        std::memcpy(&t.prbs_parity[s * 32], st, kParity);
    }
}

// ---- PN511 / PN63 and the field sync segment (reference: src/common/lfsr.h:10-26,
// src/atsc/field_sync.h:42-70,81-82).
std::vector<uint8_t> pn_sequence(unsigned state, unsigned out_bit, unsigned fb_mask, int n) {
    std::vector<uint8_t> bits(n);
    for (int i = 0; i < n; ++i) {
        bits[i] = (state >> out_bit) & 1u;
        state = (state << 1) | static_cast<unsigned>(__builtin_popcount(state & fb_mask) & 1);
    }
    return bits;
}

void build_fsync(Tables& t) {
    const auto pn511 = pn_sequence(0b10, 8, 0b110110110, 511);
    const auto pn63 = pn_sequence(0b111001, 5, 0b110000, 63);
    const char* vsb_mode = "000010100101111101011010";
    for (int odd = 0; odd < 2; ++odd) {
        auto& seg = t.fsync[odd];
        seg.assign(kSegSyms, 0);
        int n = 0;
        auto put = [&](int bit) { seg[n++] = bit ? 6 : 1; };
        put(1); put(0); put(0); put(1);
        for (uint8_t b : pn511) put(b);
        for (uint8_t b : pn63) put(b);
        for (uint8_t b : pn63) put(odd ? !b : b);  // middle PN63 inverted on odd fields
        for (uint8_t b : pn63) put(b);
        for (int i = 0; i < 24; ++i) put(vsb_mode[i] == '1');
        for (int i = 0; i < 92; ++i) put(pn63[i % 63]);
        // n == 820; the last 12 symbols are filled per field from the previous field.
    }
}

// ---- RRC taps.  The reference hard-codes 111 values + a zero (src/atsc/filter.h:71).  They are a root-raised-cosine
// pulse sampled at 2 samples per pulse period T (the 8VSB symbol rate is the Nyquist rate of the 5.38 MHz vestigial
// sideband pulse), roll-off alpha = 0.1152, 111 taps centred on tap 55, DC gain 0.1.  Evaluated here from
// the textbook closed form of the pulse with u = t/T:
//
//     h(u) = [ sin(pi u (1 - a)) + 4 a u cos(pi u (1 + a)) ] / [ pi u (1 - (4 a u)^2) ],     h(0) = 1 - a + 4a/pi
//
// (the removable singularity at 4 a u = +-1 is not hit: 4 a u = 0.2304 k never equals 1 for an integer k).  The prototype is
// rounded to single precision before it is normalised, which is what makes the result agree with the reference's constants
// to the last bit; tests/test_cabi_cpu.py pins all 112 bit patterns to tests/golden/golden.json["taps_bits"].
double rrc_pulse(double u, double a) {
    if (u == 0.0) return 1.0 - a + 4.0 * a / M_PI;
    const double s = std::sin(M_PI * u * (1.0 - a));
    const double c = std::cos(M_PI * u * (1.0 + a));
    const double w = 4.0 * a * u;
    return (s + w * c) / (M_PI * u * (1.0 - w * w));
}

void build_taps(Tables& t) {
    constexpr int kLen = 111, kCentre = kLen / 2;
    constexpr double kRollOff = 0.1152, kDcGain = 0.1, kSamplesPerPeriod = 2.0;
    float proto[kLen];
    double dc = 0.0;
    for (int n = 0; n < kLen; ++n) {
        proto[n] = static_cast<float>(rrc_pulse((n - kCentre) / kSamplesPerPeriod, kRollOff));
        dc += proto[n];
    }
    for (int n = 0; n < kLen; ++n) t.taps[n] = static_cast<float>(proto[n] * kDcGain / dc);
    t.taps[kLen] = 0.0f;   // 112th tap: pads the filter to a multiple of 16 (filter.h:71 ends in a zero as well)
    // The interpolating form's second phase (no reference counterpart: the reference runs the radio at the symbol rate,
    // util/atsc_encoder.cpp:70): the same pulse, same scale, sampled half a symbol later -- y2[2i+1] = sum_k taps_half[k] z[i+k]
    // sits between y[i] and y[i+1]; 112 taps, symmetric about k = 55.5.
    for (int n = 0; n < kLen + 1; ++n)
        t.taps_half[n] = static_cast<float>(rrc_pulse((n - kCentre - 0.5) / kSamplesPerPeriod, kRollOff) * kDcGain / dc);
}

// ---- oscillator (reference: src/atsc/offset.h:84-106).  NOT an ideal NCO: the phase is computed
// in double but rounded to float before cosf/sinf, and the per-field advance is a float complex.
void build_osc(Tables& t) {
    const double fs = 4500000.0 / 286 * 684;
    const double step = 2 * M_PI * (309411.0 - 3000000.0) / fs;
    t.osc0.resize(static_cast<size_t>(kPaddedSyms) * 2);
    for (size_t i = 0; i < static_cast<size_t>(kPaddedSyms); ++i) {
        const float phase = static_cast<float>(step * i);
        t.osc0[2 * i] = static_cast<float>(0.9 * cosf(phase));
        t.osc0[2 * i + 1] = static_cast<float>(0.9 * sinf(phase));
    }
    const float phase = static_cast<float>(step * kFieldSyms);
    t.scale[0] = cosf(phase);
    t.scale[1] = sinf(phase);
}

}  // namespace

void build(Tables& t) {
    build_prbs(t);
    build_rs(t);
    build_fsync(t);
    build_taps(t);
    build_osc(t);
}

}  // namespace atsc_tables
