/*
 * TEST INFRASTRUCTURE ONLY.  CPU restatement ("port") of the ATSC 8VSB modulator hot path of
 * philburr/atsc, written from the closed-form specification in SURVEY.md Appendix A -- not from
 * the reference's code shape.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker.  The product (atsc_amd/csrc) never links,
 * imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every stage of this file against
 *   (a) the sha256 / spot values recorded from the compiled reference in SURVEY.md Appendix B and
 *       committed under tests/golden/, and
 *   (b) when oracle/_ref/libatsc_ref.so is present, the compiled reference itself, stage by stage.
 *
 * Each function cites the reference file:line whose behaviour it restates.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PKTS 312
#define TS_BYTES 188
#define SEG_DATA 187
#define SEG_FEC 207
#define RS_PAR 20
#define FIELD_TS (PKTS * TS_BYTES)       /* 58656  */
#define FIELD_ENC (PKTS * SEG_FEC)       /* 64584  */
#define SEG_SYMS 832
#define FIELD_SYMS (313 * SEG_SYMS)      /* 260416 */
#define PAD_SYMS (FIELD_SYMS + SEG_SYMS) /* 261248 */
#define IL_DEPTH (208 * 51)              /* 10608: farthest look-back of the interleaver */
#define NTAPS 112

/* ---------------------------------------------------------------- static tables */
static uint8_t g_prbs[SEG_DATA * PKTS];
static uint8_t g_gf_exp[512], g_gf_log[256];
static uint8_t g_rs_gen[RS_PAR + 1]; /* g[0..20], g[20]=1, ascending powers */
static uint8_t g_fsync[2][SEG_SYMS - 12];
static float g_taps[NTAPS];
static float g_osc0[PAD_SYMS * 2];
static float g_scale[2];
static int g_init_done;

/* PRBS: src/atsc/randomize.h:13-33 (state 0xf180, taps 0x9c65, 8 output bits picked per byte). */
static void init_prbs(void) {
    uint32_t st = 0xf180;
    for (int i = 0; i < SEG_DATA * PKTS; i++) {
        g_prbs[i] = (uint8_t)(((st & 0x3c00) >> 6) | ((st & 0x40) >> 3) | ((st & 0xc) >> 1) | (st & 1));
        st <<= 1;
        if (st & 0x10000) st ^= (0x9c65u << 1) | 1;
    }
}

/* GF(2^8)/0x11d antilog/log: src/atsc/galois.h:85-113.  Generator polynomial with roots
 * alpha^0..alpha^19: src/atsc/reed_solomon.h:23-39. */
static uint8_t gf_mul(uint8_t a, uint8_t b) {
    if (!a || !b) return 0;
    return g_gf_exp[g_gf_log[a] + g_gf_log[b]];
}
static void init_rs(void) {
    unsigned v = 1;
    for (int i = 0; i < 255; i++) {
        g_gf_exp[i] = g_gf_exp[i + 255] = (uint8_t)v;
        g_gf_log[v] = (uint8_t)i;
        v <<= 1;
        if (v & 0x100) v ^= 0x11d;
    }
    memset(g_rs_gen, 0, sizeof g_rs_gen);
    g_rs_gen[0] = 1;
    for (int i = 0; i < RS_PAR; i++) { /* multiply by (x + alpha^i) */
        uint8_t root = g_gf_exp[i];
        for (int j = i + 1; j > 0; j--) g_rs_gen[j] = g_rs_gen[j - 1] ^ gf_mul(g_rs_gen[j], root);
        g_rs_gen[0] = gf_mul(g_rs_gen[0], root);
    }
}

/* PN sequences: src/common/lfsr.h:10-26 (Fibonacci LFSR, output tap SHIFT_OUT, feedback = parity
 * of state&MASK).  Field sync layout: src/atsc/field_sync.h:42-70. */
static void lfsr_bits(unsigned state, unsigned out_shift, unsigned mask, int n, uint8_t* out) {
    for (int i = 0; i < n; i++) {
        out[i] = (state >> out_shift) & 1;
        unsigned u = state & mask;
        state = (state << 1) | (unsigned)__builtin_parity(u);
    }
}
static void init_fsync(void) {
    uint8_t pn511[511], pn63[63];
    static const uint8_t vsb[24] = {0, 0, 0, 0, 1, 0, 1, 0, 0, 1, 0, 1, 1, 1, 1, 1, 0, 1, 0, 1, 1, 0, 1, 0};
    lfsr_bits(0x2, 8, 0x1b6, 511, pn511);
    lfsr_bits(0x39, 5, 0x30, 63, pn63);
    for (int odd = 0; odd < 2; odd++) {
        uint8_t* t = g_fsync[odd];
        int n = 0;
        t[n++] = 6; t[n++] = 1; t[n++] = 1; t[n++] = 6;
        for (int i = 0; i < 511; i++) t[n++] = pn511[i] ? 6 : 1;
        for (int i = 0; i < 63; i++) t[n++] = pn63[i] ? 6 : 1;
        for (int i = 0; i < 63; i++) t[n++] = ((pn63[i] != 0) == (odd == 0)) ? 6 : 1; /* inverted on odd fields */
        for (int i = 0; i < 63; i++) t[n++] = pn63[i] ? 6 : 1;
        for (int i = 0; i < 24; i++) t[n++] = vsb[i] ? 6 : 1;
        for (int i = 0; i < 92; i++) t[n++] = pn63[i % 63] ? 6 : 1;
    }
}

/* RRC taps.  The reference holds them as 111 literal constants and a closing zero (src/atsc/filter.h:71); they are the
 * root-raised-cosine pulse h(u) = [sin(pi u (1-a)) + 4 a u cos(pi u (1+a))] / [pi u (1 - (4 a u)^2)], h(0) = 1 - a + 4a/pi,
 * at u = (n - 55)/2, roll-off a = 0.1152, rounded to float and then scaled to a DC gain of 0.1.  Evaluated
 * from that textbook form; tests/test_oracle.py pins all 112 bit patterns to tests/golden/golden.json["taps_bits"], which
 * were read out of the compiled reference. */
static double rrc_at(double u, double a) {
    if (u == 0.0) return 1.0 - a + 4.0 * a / M_PI;
    double w = 4.0 * a * u; /* never +-1 here: 0.2304 k is not 1 for an integer k */
    return (sin(M_PI * u * (1.0 - a)) + w * cos(M_PI * u * (1.0 + a))) / (M_PI * u * (1.0 - w * w));
}

static void init_taps(void) {
    float shape[111];
    double total = 0;
    for (int n = 0; n < 111; n++) {
        shape[n] = (float)rrc_at((n - 55) / 2.0, 0.1152);
        total += shape[n];
    }
    for (int n = 0; n < 111; n++) g_taps[n] = (float)(shape[n] * 0.1 / total);
    g_taps[111] = 0.0f;
}

/* Oscillator table and per-field rotation: src/atsc/offset.h:84-106.  The phase argument is
 * computed in double and handed to cosf/sinf, i.e. rounded to float first. */
static void init_osc(void) {
    const double fs = 4500000.0 / 286 * 684;
    const double inc = 2 * M_PI * (309411.0 - 3000000.0) / fs;
    for (size_t i = 0; i < PAD_SYMS; i++) {
        float a = (float)(inc * (double)i);
        g_osc0[2 * i] = (float)(0.9 * cosf(a));
        g_osc0[2 * i + 1] = (float)(0.9 * sinf(a));
    }
    float a = (float)(inc * (double)FIELD_SYMS);
    g_scale[0] = cosf(a);
    g_scale[1] = sinf(a);
}

void oracle_init(void) {
    if (g_init_done) return;
    init_prbs();
    init_rs();
    init_fsync();
    init_taps();
    init_osc();
    g_init_done = 1;
}

/* table taps for tests */
const uint8_t* oracle_prbs(void) { oracle_init(); return g_prbs; }
const uint8_t* oracle_rs_gen(void) { oracle_init(); return g_rs_gen; }
const uint8_t* oracle_fsync(int odd) { oracle_init(); return g_fsync[odd & 1]; }
const float* oracle_taps(void) { oracle_init(); return g_taps; }
const float* oracle_osc0(void) { oracle_init(); return g_osc0; }
const float* oracle_scale(void) { oracle_init(); return g_scale; }

/* ---------------------------------------------------------------- per-stream state */
typedef struct {
    uint8_t hist[IL_DEPTH + FIELD_ENC]; /* coded-byte stream: [previous tail | current field] */
    uint8_t dz2[12], d1[12], d2[12];    /* precoder delay, conv. encoder delays (trellis.h:12-34) */
    uint8_t reserved[12];               /* last 12 symbols of the previous field (field_sync.h:17,23) */
    uint64_t field_index;
    float* osc;                         /* this stream's oscillator table (offset.h:9) */
    float* padded;                      /* mixed, padded field (261248 complex) */
} oracle_t;

void* oracle_create(void) {
    oracle_init();
    oracle_t* o = (oracle_t*)calloc(1, sizeof *o);
    o->osc = (float*)malloc(sizeof(float) * 2 * PAD_SYMS);
    o->padded = (float*)malloc(sizeof(float) * 2 * PAD_SYMS);
    memcpy(o->osc, g_osc0, sizeof(float) * 2 * PAD_SYMS);
    return o;
}
void oracle_destroy(void* h) {
    oracle_t* o = (oracle_t*)h;
    free(o->osc);
    free(o->padded);
    free(o);
}

/* ---------------------------------------------------------------- stages */

/* a2: src/atsc/randomize.h:43-54.  Sync byte dropped unchecked; parity area zeroed. */
void oracle_randomize(const uint8_t* ts, uint8_t* enc) {
    oracle_init();
    for (int s = 0; s < PKTS; s++) {
        for (int k = 0; k < SEG_DATA; k++) enc[s * SEG_FEC + k] = ts[s * TS_BYTES + 1 + k] ^ g_prbs[s * SEG_DATA + k];
        memset(enc + s * SEG_FEC + SEG_DATA, 0, RS_PAR);
    }
}

/* a3: src/atsc/reed_solomon.h:192-209, 230-242.  Systematic RS(207,187): parity = remainder of
 * d(x)*x^20 by g(x), highest-order remainder coefficient first. */
void oracle_rs_parity(const uint8_t* data187, uint8_t* par20) {
    uint8_t bb[RS_PAR];
    memset(bb, 0, sizeof bb);
    for (int k = 0; k < SEG_DATA; k++) {
        uint8_t fb = data187[k] ^ bb[0];
        for (int j = 0; j < RS_PAR - 1; j++) bb[j] = bb[j + 1] ^ gf_mul(g_rs_gen[RS_PAR - 1 - j], fb);
        bb[RS_PAR - 1] = gf_mul(g_rs_gen[0], fb);
    }
    memcpy(par20, bb, RS_PAR);
}
void oracle_rs_field(uint8_t* enc) {
    oracle_init();
    for (int s = 0; s < PKTS; s++) oracle_rs_parity(enc + s * SEG_FEC, enc + s * SEG_FEC + SEG_DATA);
}

/* a4: src/atsc/interleaver.h:22-108 as the closed form of SURVEY.md 8(a4):
 *   IL[M] = E[M - 208*(M mod 52)] on the continuous coded stream, zeros before the first field.
 * `hist` holds [10608 B tail of the past | 64584 B of this field]; il gets this field's 64584 B in
 * natural (transmission) order.  The reference's in-memory layout additionally applies the
 * trellis demux transpose (interleaver.h:55-60): oracle_il_to_ref_layout() reproduces it for
 * comparison with the reference's `current_field` buffer. */
static void interleave_field(const uint8_t* hist, uint8_t* il) {
    for (int m = 0; m < FIELD_ENC; m++) il[m] = hist[IL_DEPTH + m - 208 * (m % 52)];
}
static int trellis_shift(int g) { return 4 * ((12 * g / SEG_FEC) % 3); }
void oracle_il_to_ref_layout(const uint8_t* il, uint8_t* ref) {
    for (int m = 0; m < FIELD_ENC; m++) {
        int g = m / 12, e = (m + trellis_shift(g)) % 12;
        ref[(g / 13) * 156 + e * 13 + (g % 13)] = il[m];
    }
}

/* a5+a6: src/atsc/trellis.h:10-51 (scalar spec), :549-576 (output commutator),
 * src/atsc/field_sync.h:12-26, src/atsc/signal.h:13-18.  Emits u8 level indices 0..7. */
static void trellis_sync_field(oracle_t* o, const uint8_t* il, uint8_t* sym) {
    /* per-encoder symbol streams, 21528 symbols each */
    static uint8_t v[12][FIELD_ENC / 12 * 4];
    for (int g = 0; g < FIELD_ENC / 12; g++) {
        int sh = trellis_shift(g);
        for (int e = 0; e < 12; e++) {
            uint8_t b = il[12 * g + ((e - sh + 12) % 12)];
            for (int p = 0; p < 4; p++) {
                int x2 = (b >> (7 - 2 * p)) & 1, x1 = (b >> (6 - 2 * p)) & 1;
                o->dz2[e] ^= (uint8_t)x2;
                uint8_t z0 = o->d1[e];
                o->d1[e] = (uint8_t)x1 ^ o->d2[e];
                o->d2[e] = z0;
                v[e][4 * g + p] = (uint8_t)(4 * o->dz2[e] + 2 * x1 + z0);
            }
        }
    }
    const uint8_t* fs = g_fsync[o->field_index & 1];
    memcpy(sym, fs, SEG_SYMS - 12);
    memcpy(sym + SEG_SYMS - 12, o->reserved, 12);
    for (int s = 0; s < PKTS; s++) {
        uint8_t* seg = sym + (s + 1) * SEG_SYMS;
        seg[0] = 6; seg[1] = 1; seg[2] = 1; seg[3] = 6;
        for (int j = 0; j < 828; j++) seg[4 + j] = v[(4 * s + j) % 12][69 * s + j / 12];
    }
    memcpy(o->reserved, sym + FIELD_SYMS - 12, 12);
}

/* a7: src/atsc/offset.h:11-45 (FMA path).  The input is real, so each product is a single
 * rounded multiply (fma(x,c,-(0*ci)) == x*c).  Table advance: re=fma(cr,er,-(ci*ei)),
 * im=fma(ci,er,cr*ei), applied to all 261248 entries. */
static void mix_field(oracle_t* o, const uint8_t* sym) {
    float* X = o->padded;
    float* T = o->osc;
    const float er = g_scale[0], ei = g_scale[1];
    for (int i = 0; i < PAD_SYMS; i++) {
        float x;
        if (i < FIELD_SYMS) x = (float)(2 * sym[i] - 7) + 1.25f;
        else if (i < FIELD_SYMS + SEG_SYMS - 12) x = (float)(2 * sym[i - FIELD_SYMS] - 7) + 1.25f; /* field_sync.h:16 */
        else x = 0.0f;
        float cr = T[2 * i], ci = T[2 * i + 1];
        X[2 * i] = x * cr;
        X[2 * i + 1] = x * ci;
        float t2 = ci * ei, t3 = cr * ei;
        T[2 * i] = fmaf(cr, er, -t2);
        T[2 * i + 1] = fmaf(ci, er, t3);
    }
}

/* a8: src/atsc/filter.h:16-69 (FMA path): 16 accumulator chains (chain q takes taps 16i+q in
 * order), then ((v0+v1)+v2)+v3 lane-wise, then r0 + ((r1+r2)+r3). */
static void fir_field(const oracle_t* o, float* iq) {
    const float* X = o->padded;
    for (int i = 0; i < FIELD_SYMS; i++) {
        float acc[16][2];
        memset(acc, 0, sizeof acc);
        const float* x = X + 2 * (size_t)i;
        for (int blk = 0; blk < NTAPS / 16; blk++)
            for (int q = 0; q < 16; q++) {
                float c = g_taps[16 * blk + q];
                acc[q][0] = fmaf(x[2 * (16 * blk + q)], c, acc[q][0]);
                acc[q][1] = fmaf(x[2 * (16 * blk + q) + 1], c, acc[q][1]);
            }
        float r[4][2];
        for (int k = 0; k < 4; k++)
            for (int c = 0; c < 2; c++) r[k][c] = ((acc[k][c] + acc[4 + k][c]) + acc[8 + k][c]) + acc[12 + k][c];
        iq[2 * (size_t)i] = r[0][0] + ((r[1][0] + r[2][0]) + r[3][0]);
        iq[2 * (size_t)i + 1] = r[0][1] + ((r[1][1] + r[2][1]) + r[3][1]);
    }
}

/* ---------------------------------------------------------------- field driver
 * Mirrors atsc_encoder_impl::process_field (src/atsc/atsc.cpp:79-90) over whole fields; every
 * out pointer may be NULL.  il is in natural order (see interleave_field). */
void oracle_process_fields(void* h, const uint8_t* ts, int nfields, uint8_t* enc, uint8_t* il, uint8_t* sym,
                           float* iq) {
    oracle_t* o = (oracle_t*)h;
    uint8_t* ilbuf = (uint8_t*)malloc(FIELD_ENC);
    uint8_t* symbuf = (uint8_t*)malloc(FIELD_SYMS);
    for (int f = 0; f < nfields; f++) {
        uint8_t* cur = o->hist + IL_DEPTH;
        oracle_randomize(ts + (size_t)f * FIELD_TS, cur);
        oracle_rs_field(cur);
        if (enc) memcpy(enc + (size_t)f * FIELD_ENC, cur, FIELD_ENC);
        interleave_field(o->hist, ilbuf);
        if (il) memcpy(il + (size_t)f * FIELD_ENC, ilbuf, FIELD_ENC);
        trellis_sync_field(o, ilbuf, symbuf);
        if (sym) memcpy(sym + (size_t)f * FIELD_SYMS, symbuf, FIELD_SYMS);
        if (iq) {
            mix_field(o, symbuf);
            fir_field(o, iq + (size_t)f * FIELD_SYMS * 2);
        } else {
            /* keep the oscillator in step even when IQ is not requested */
            mix_field(o, symbuf);
        }
        memmove(o->hist, o->hist + FIELD_ENC, IL_DEPTH);
        o->field_index++;
    }
    free(ilbuf);
    free(symbuf);
}

/* Symbols only, oscillator untouched (fast path for large symbol-parity runs). */
void oracle_symbols(void* h, const uint8_t* ts, int nfields, uint8_t* sym) {
    oracle_t* o = (oracle_t*)h;
    uint8_t* ilbuf = (uint8_t*)malloc(FIELD_ENC);
    for (int f = 0; f < nfields; f++) {
        uint8_t* cur = o->hist + IL_DEPTH;
        oracle_randomize(ts + (size_t)f * FIELD_TS, cur);
        oracle_rs_field(cur);
        interleave_field(o->hist, ilbuf);
        trellis_sync_field(o, ilbuf, sym + (size_t)f * FIELD_SYMS);
        memmove(o->hist, o->hist + FIELD_ENC, IL_DEPTH);
        o->field_index++;
    }
    free(ilbuf);
}

/* Mixer + FIR only, from given symbols (used to check the GPU FIR in isolation). */
void oracle_mix_fir(void* h, const uint8_t* sym, int nfields, float* iq) {
    oracle_t* o = (oracle_t*)h;
    for (int f = 0; f < nfields; f++) {
        mix_field(o, sym + (size_t)f * FIELD_SYMS);
        fir_field(o, iq + (size_t)f * FIELD_SYMS * 2);
    }
}

/* Trellis state in the reference's get_carry() convention (trellis.h:85-97, :240-242):
 * z2 = precoder delay, z0a = Z0 of the last odd-time symbol ... expressed from (D1,D2). */
void oracle_get_state(void* h, uint8_t* z2, uint8_t* d1, uint8_t* d2, uint8_t* reserved, uint64_t* field_index) {
    oracle_t* o = (oracle_t*)h;
    if (z2) memcpy(z2, o->dz2, 12);
    if (d1) memcpy(d1, o->d1, 12);
    if (d2) memcpy(d2, o->d2, 12);
    if (reserved) memcpy(reserved, o->reserved, 12);
    if (field_index) *field_index = o->field_index;
}
const float* oracle_osc_table(void* h) { return ((oracle_t*)h)->osc; }
