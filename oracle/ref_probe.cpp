// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or called from the product path.
//
// C-ABI probe around the *compiled reference* (philburr/atsc).  This file contains no reference
// code: it pulls the reference translation unit in by path at build time
// (`-I/root/reference/src`, see oracle/Makefile) and is built only where /root/reference exists.
// The resulting oracle/_ref/libatsc_ref.so is git-ignored; it travels to the GPU box as a
// prebuilt binary and is used (a) to pin the C restatement in oracle/atsc_oracle.c and (b) as the
// "reference" CPU baseline in bench.py.
//
// Stage hooks follow atsc_encoder_impl::process_field (src/atsc/atsc.cpp:79-90); private members
// are reached with -fno-access-control (SURVEY.md Appendix B recipe).
#include <cstdint>
#include <cstring>
#include <cmath>
#include "atsc/atsc.cpp"   // resolved through -I/root/reference/src

extern "C" {

struct ref_handle {
    atsc_encoder_impl enc;
};

void* ref_create() { return new ref_handle(); }
void ref_destroy(void* h) { delete static_cast<ref_handle*>(h); }

// Public-API path: atsc_encoder::process (src/atsc/atsc.cpp:59-77).  Every completed field is
// appended to iq_out (2*260416 floats each); returns the number of callbacks fired.
int ref_process(void* h, const uint8_t* pkts, unsigned count, float* iq_out, int max_fields) {
    auto* r = static_cast<ref_handle*>(h);
    int fired = 0;
    atsc::atsc_encoder* e = &r->enc;
    e->process(const_cast<uint8_t*>(pkts), count, [&](void* p, unsigned bytes) {
        if (iq_out && fired < max_fields) memcpy((char*)iq_out + (size_t)fired * bytes, p, bytes);
        fired++;
    });
    return fired;
}

// Stage-by-stage path over whole fields with taps after each stage (all out pointers nullable):
//   enc : 64584 B/field after randomizer+RS          (atsc.cpp:80-81)
//   il  : 64584 B/field, reference interleaver layout (atsc.cpp:82, before the swap)
//   sym : 260416 B/field, u8 level index after field_sync (atsc.cpp:84)
//   mix : 261248 complex64/field after the mixer      (atsc.cpp:85)
//   iq  : 260416 complex64/field after the FIR        (atsc.cpp:86)
void ref_process_fields(void* h, const uint8_t* ts, int nfields, uint8_t* enc, uint8_t* il,
                        uint8_t* sym, float* mix, float* iq) {
    auto* r = static_cast<ref_handle*>(h);
    auto& E = r->enc;
    for (int f = 0; f < nfields; f++) {
        memcpy(E.input->data(), ts + (size_t)f * 312 * 188, 312 * 188);
        E.randomizer.randomize_pkts(*E.encoded, *E.input);
        E.fec.process_field(*E.encoded);
        if (enc) memcpy(enc + (size_t)f * 64584, E.encoded->data(), 64584);
        E.interleaver.process(E.current_field, E.next_field, *E.encoded);
        if (il) memcpy(il + (size_t)f * 64584, E.current_field.data(), 64584);
        E.trellis.process(*E.out, E.current_field);
        E.field_sync.process_field(*E.out, E.saved_symbols);
        if (sym) {
            for (int i = 0; i < 260416; i++)
                sym[(size_t)f * 260416 + i] = (uint8_t)lrintf(((*E.out)[i].real() - 1.25f + 7.0f) / 2.0f);
        }
        E.offset.process_field(*E.out);
        if (mix) memcpy(mix + (size_t)f * 261248 * 2, E.out->data(), 261248 * 8);
        E.filter.process_field(*E.filtered, *E.out);
        if (iq) memcpy(iq + (size_t)f * 260416 * 2, E.filtered->data(), 260416 * 8);
        std::swap(E.current_field, E.next_field);
    }
}

// Trellis carries z2/z0a/z0b for encoders 0..11 (trellis.h:85-97) -> 36 bytes of 0/1.
void ref_get_carry(void* h, uint8_t* out36) {
    auto c = static_cast<ref_handle*>(h)->enc.trellis.get_carry();
    for (int k = 0; k < 3; k++)
        for (int e = 0; e < 12; e++) out36[k * 12 + e] = c[k][e];
}

// Current oscillator table of this instance (offset.h:9,81) and the per-field scale (offset.h:108).
void ref_osc_table(void* h, float* out /*261248*2*/) {
    auto* r = static_cast<ref_handle*>(h);
    memcpy(out, r->enc.offset._table->data(), 261248 * 8);
}
void ref_osc_scale(float* out2) {
    out2[0] = atsc_offset::scale[0].real();
    out2[1] = atsc_offset::scale[0].imag();
}
// The 112 hard-coded FIR coefficients (filter.h:71).
void ref_taps(float* out112) { memcpy(out112, atsc_rrc_filter::coefficients, 112 * sizeof(float)); }

// Static tables, for pinning the restatement's generators.
void ref_prbs(uint8_t* out /*58344*/) {
    for (int i = 0; i < 187 * 312; i++) out[i] = (uint8_t)atsc_randomize<void>::table[i];
}
void ref_interleaver_table(uint32_t* out /*64584*/) {
    memcpy(out, atsc_interleaver<true>::table_.data(), 64584 * 4);
}

}  // extern "C"
