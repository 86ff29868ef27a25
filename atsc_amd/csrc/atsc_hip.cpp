// C-ABI host layer of the MI355X 8VSB modulator (see include/atsc_hip.h).
// Mirrors atsc_encoder_impl (reference: src/atsc/atsc.cpp:15-91): packet buffering, field
// sequencing, per-field callbacks -- with the seven stages replaced by batched gfx950 kernels.
#include "atsc_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "kernels.h"
#include "tables.h"

using namespace atsc_k;

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) return static_cast<int>(_e); \
    } while (0)

namespace {
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
}  // namespace

struct atsc_hip_encoder {
    atsc_hip_config cfg{};
    atsc_tables::Tables tab;
    DeviceTables dtab{};
    Buffers buf{};
    uint8_t* d_sym = nullptr;   // internal symbol buffer [channels][max_fields][260416]
    size_t sym_stride = 0;
    float2* d_osc[2] = {nullptr, nullptr};
    float2* d_osc0 = nullptr;     // T_0, read-only after create (atsc_hip_seek)
    float2* d_osc_tab = nullptr;  // [max_fields][261248] oscillator table of the current batch
    int osc_cur = 0;
    float2 scale{};
    uint64_t field_index = 0;
    int pending_fields = 0;    // fields of a batch between encode_begin and encode_finish
    int last_fields = 0;       // field count of the last batch (internal symbol buffer is packed with it)
    bool sym_internal = false; // the last batch wrote its symbols to the internal buffer (no caller buffer, not the fused form)
    int iq_format = ATSC_HIP_IQ_CF32;
    int fir_form = ATSC_HIP_FIR_AUTO;
    int upsample = 1;          // output samples per symbol (atsc_hip_set_upsample)
    void* d_tables = nullptr;

    // streaming path (1 channel): two slots so that the upload and the kernels of batch k+1 run while the IQ of batch k
    // comes back over the host link and its callbacks run
    std::vector<uint8_t> pending;        // bytes of an incomplete field + queued whole fields
    struct Slot {
        uint8_t* h_ts = nullptr;         // pinned
        uint8_t* d_ts = nullptr;
        uint8_t* d_iq = nullptr;
        uint8_t* h_iq = nullptr;         // pinned: what the callbacks see
        hipEvent_t encoded = nullptr;    // kernels of the slot's batch done (compute stream)
        hipEvent_t copied = nullptr;     // IQ of the slot's batch in h_iq (copy stream)
    } slot[2];
    hipStream_t compute_stream = nullptr, copy_stream = nullptr;
    // multi-stream front (atsc_hip_process_multi): one FIFO and one callback per channel, two slots of channels x max_fields
    struct Multi {
        std::vector<std::vector<uint8_t>> fifo;      // per channel: TS bytes not yet encoded
        std::vector<atsc_hip_field_cb> cb;           // per channel: the callback of its latest call
        std::vector<void*> user;
        Slot slot[2];
        size_t slot_iq_bytes = 0;                    // what the slots' IQ buffers were sized for
    } multi;
    hipStream_t last_stream = nullptr;   // the stream the handle's state was last touched on (batch calls, seek)
    hipEvent_t order_ev = nullptr;       // orders atsc_hip_process's private streams behind it

    // timing: a ring of event sets, one per batch, so a timed region can be read back afterwards
    struct EventSet {
        hipEvent_t ev[ATSC_HIP_NSTAGES + 1] = {};
        bool recorded[ATSC_HIP_NSTAGES + 1] = {};
    };
    std::vector<EventSet> slots;
    bool ev_valid = false;
    unsigned mark_mask = ~0u;   // which marks a batch records (atsc_hip_timing_select)
    size_t batches = 0;         // batches launched since the last timing reset
    EventSet* cur = nullptr;
};

extern "C" const char* atsc_hip_version(void) { return "atsc_hip 0.1 (gfx950)"; }

extern "C" const char* atsc_hip_error_string(int code) {
    switch (code) {
        case ATSC_HIP_OK: return "ok";
        case ATSC_HIP_ERR_ARG: return "invalid argument";
        case ATSC_HIP_ERR_NO_DEVICE: return "no HIP device: the 8VSB path has no CPU fallback";
        case ATSC_HIP_ERR_TOO_LARGE: return "batch larger than the handle was created for";
        case ATSC_HIP_ERR_STATE: return "call not valid in this state";
        default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "unknown error";
    }
}

static int destroy_impl(atsc_hip_encoder* e) {
    if (!e) return 0;
    hipSetDevice(e->cfg.device);
    hipFree(e->buf.coded); hipFree(std::min(e->buf.tail, e->buf.tail_next)); hipFree(e->buf.parity); hipFree(e->buf.il); hipFree(e->buf.par); hipFree(e->buf.cin);
    hipFree(e->buf.carry); hipFree(e->buf.span_par);
    hipFree(std::min(e->buf.last_group, e->buf.last_group_next));   // one allocation, two halves that swap roles
    hipFree(e->d_sym); hipFree(e->d_osc[0]); hipFree(e->d_osc[1]); hipFree(e->d_osc0); hipFree(e->d_osc_tab); hipFree(e->d_tables);
    for (auto& sl : e->slot) {
        hipFree(sl.d_ts); hipFree(sl.d_iq);
        if (sl.h_ts) hipHostFree(sl.h_ts);
        if (sl.h_iq) hipHostFree(sl.h_iq);
        if (sl.encoded) hipEventDestroy(sl.encoded);
        if (sl.copied) hipEventDestroy(sl.copied);
    }
    for (auto& sl : e->multi.slot) {
        hipFree(sl.d_ts); hipFree(sl.d_iq);
        if (sl.h_ts) hipHostFree(sl.h_ts);
        if (sl.h_iq) hipHostFree(sl.h_iq);
        if (sl.encoded) hipEventDestroy(sl.encoded);
        if (sl.copied) hipEventDestroy(sl.copied);
    }
    if (e->order_ev) hipEventDestroy(e->order_ev);
    if (e->compute_stream) hipStreamDestroy(e->compute_stream);
    if (e->copy_stream) hipStreamDestroy(e->copy_stream);
    for (auto& set : e->slots)
        for (auto& ev : set.ev)
            if (ev) hipEventDestroy(ev);
    delete e;
    return 0;
}

extern "C" void atsc_hip_destroy(atsc_hip_encoder* enc) { destroy_impl(enc); }

static int create_impl(const atsc_hip_config* cfg, atsc_hip_encoder** out);

extern "C" int atsc_hip_create(const atsc_hip_config* cfg, atsc_hip_encoder** out) {
    if (!cfg || !out || cfg->channels < 1 || cfg->max_fields < 1) return ATSC_HIP_ERR_ARG;
    *out = nullptr;
    // the table generators and the staging buffer allocate host memory: no C++ exception may cross the C boundary
    try {
        return create_impl(cfg, out);
    } catch (const std::bad_alloc&) {
        return static_cast<int>(hipErrorOutOfMemory);
    } catch (...) {
        return ATSC_HIP_ERR_ARG;
    }
}

namespace {
// frees a half-built handle unless release() was called
struct HandleGuard {
    atsc_hip_encoder* e;
    ~HandleGuard() { if (e) destroy_impl(e); }
    atsc_hip_encoder* release() { atsc_hip_encoder* r = e; e = nullptr; return r; }
};
}  // namespace

static int create_impl(const atsc_hip_config* cfg, atsc_hip_encoder** out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ATSC_HIP_ERR_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return ATSC_HIP_ERR_ARG;
    HIP_TRY(hipSetDevice(cfg->device));

    auto* e = new (std::nothrow) atsc_hip_encoder();
    if (!e) return static_cast<int>(hipErrorOutOfMemory);
    HandleGuard guard{e};   // an exception (std::bad_alloc from the table build) or an early return frees the handle
    e->cfg = *cfg;
    atsc_tables::build(e->tab);
    const size_t C = cfg->channels, F = cfg->max_fields;

#define TRY_OR_FREE(expr)                                \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return static_cast<int>(_e);  \
    } while (0)

    // constant tables: one allocation [prbs188 | rs_rows | prbs parity | fsync even | fsync odd | prbs | prbs rows]
    const size_t off_prbs = 0, off_rows = align_up(kFieldTs, 256), off_pp = off_rows + atsc_tables::kRsLdsBytes, off_fs = off_pp + kPackets * 32,
                 off_pr = align_up(off_fs + 2 * kSegSyms, 256);
    const size_t off_prr = align_up(off_pr + e->tab.prbs.size(), 256);
    const size_t tab_bytes = off_prr + kPackets * 208;
    TRY_OR_FREE(hipMalloc(&e->d_tables, tab_bytes));
    {
        std::vector<uint8_t> h(tab_bytes, 0);
        std::memcpy(h.data() + off_prbs, e->tab.prbs188.data(), kFieldTs);
        std::memcpy(h.data() + off_rows, e->tab.rs_lds.data(), atsc_tables::kRsLdsBytes);
        std::memcpy(h.data() + off_pp, e->tab.prbs_parity.data(), kPackets * 32);
        std::memcpy(h.data() + off_fs, e->tab.fsync[0].data(), kSegSyms);
        std::memcpy(h.data() + off_fs + kSegSyms, e->tab.fsync[1].data(), kSegSyms);
        std::memcpy(h.data() + off_pr, e->tab.prbs.data(), e->tab.prbs.size());
        for (int q = 0; q < kPackets; ++q) std::memcpy(h.data() + off_prr + q * 208, e->tab.prbs.data() + q * 187, 187);
        TRY_OR_FREE(hipMemcpy(e->d_tables, h.data(), tab_bytes, hipMemcpyHostToDevice));
    }
    auto* tb = static_cast<uint8_t*>(e->d_tables);
    e->dtab.prbs188 = reinterpret_cast<const uint32_t*>(tb + off_prbs);
    e->dtab.rs_lds = reinterpret_cast<const uint4*>(tb + off_rows);
    e->dtab.prbs_parity = reinterpret_cast<const uint4*>(tb + off_pp);
    e->dtab.fsync = tb + off_fs;
    e->dtab.prbs = tb + off_pr;
    e->dtab.prbs_rows = tb + off_prr;
    TRY_OR_FREE(upload_fft_tables(e->tab.taps));
    TRY_OR_FREE(upload_mm_tables(e->tab.taps, e->tab.taps_half));
    TRY_OR_FREE(upload_w_tables(e->tab.taps));

    // per-channel working buffers
    e->buf.max_fields = cfg->max_fields;
    e->buf.coded_stride = align_up(kTail + F * kFieldCoded, 256);
    e->buf.il_stride = align_up(F * kFieldCoded, 256);
    e->sym_stride = F * kFieldSyms;
    if (cfg->flags & ATSC_HIP_FLAG_TAPS) TRY_OR_FREE(hipMalloc(&e->buf.coded, C * e->buf.coded_stride));
    TRY_OR_FREE(hipMalloc(&e->buf.tail, 2 * C * kTail));   // current | next (swapped behind every interleaver launch)
    e->buf.tail_next = e->buf.tail + C * kTail;
    TRY_OR_FREE(hipMalloc(&e->buf.parity, C * F * kPackets * 32));
    TRY_OR_FREE(hipMalloc(&e->buf.il, C * e->buf.il_stride));
    TRY_OR_FREE(hipMalloc(&e->buf.par, C * F * kParts * sizeof(uint4)));
    TRY_OR_FREE(hipMalloc(&e->buf.cin, C * F * kParts * sizeof(uint4)));
    TRY_OR_FREE(hipMalloc(&e->buf.carry, C * sizeof(uint4)));
    TRY_OR_FREE(hipMalloc(&e->buf.span_par, C * sizeof(uint4)));
    TRY_OR_FREE(hipMalloc(&e->buf.last_group, 2 * C * 16));   // current | next (the host swaps them after every batch)
    e->buf.last_group_next = e->buf.last_group + C * 16;
    // (the internal symbol buffer [C][F][260416] is allocated by the first batch that needs it: the fused form never does)
    // zero history (reference: atsc.cpp:24-26,34; trellis.h:55-59)
    if (e->buf.coded) TRY_OR_FREE(hipMemset(e->buf.coded, 0, C * e->buf.coded_stride));
    TRY_OR_FREE(hipMemset(e->buf.tail, 0, 2 * C * kTail));
    TRY_OR_FREE(hipMemset(e->buf.carry, 0, C * sizeof(uint4)));
    TRY_OR_FREE(hipMemset(e->buf.span_par, 0, C * sizeof(uint4)));
    TRY_OR_FREE(hipMemset(e->buf.last_group, 0, 2 * C * 16));

    // oscillator state (double-buffered), T_0 and the per-field rotation
    for (int i = 0; i < 2; ++i) TRY_OR_FREE(hipMalloc(&e->d_osc[i], sizeof(float2) * kPaddedSyms));
    TRY_OR_FREE(hipMemcpy(e->d_osc[0], e->tab.osc0.data(), sizeof(float2) * kPaddedSyms, hipMemcpyHostToDevice));
    TRY_OR_FREE(hipMalloc(&e->d_osc0, sizeof(float2) * kPaddedSyms));
    TRY_OR_FREE(hipMemcpy(e->d_osc0, e->tab.osc0.data(), sizeof(float2) * kPaddedSyms, hipMemcpyHostToDevice));
    TRY_OR_FREE(hipMalloc(&e->d_osc_tab, sizeof(float2) * (kPaddedSyms * F + kOscTabSlack)));
    e->scale = make_float2(e->tab.scale[0], e->tab.scale[1]);

    if (cfg->flags & ATSC_HIP_FLAG_TIMING) {
        e->slots.resize(1);
        for (auto& ev : e->slots[0].ev) TRY_OR_FREE(hipEventCreate(&ev));
        e->ev_valid = true;
    }
    TRY_OR_FREE(hipDeviceSynchronize());
#undef TRY_OR_FREE
    *out = guard.release();
    return ATSC_HIP_OK;
}

namespace {

void mark(atsc_hip_encoder* e, hipStream_t s, int idx) {
    if (e->cur && (e->mark_mask >> idx & 1u)) {
        hipEventRecord(e->cur->ev[idx], s);
        e->cur->recorded[idx] = true;
    }
}

// picks the event set of this batch (slot = batch number modulo the reserved ring)
void begin_batch_timing(atsc_hip_encoder* e) {
    e->cur = nullptr;
    if (!e->ev_valid) return;
    e->cur = &e->slots[e->batches % e->slots.size()];
    std::fill(std::begin(e->cur->recorded), std::end(e->cur->recorded), false);
    e->batches++;
}

int read_timings(atsc_hip_encoder* e, const atsc_hip_encoder::EventSet& set, float* ms) {
    for (int i = 0; i < ATSC_HIP_NSTAGES; ++i) ms[i] = 0.f;
    int prev = -1;  // each stage runs from its own mark to the next recorded mark
    for (int i = 0; i <= ATSC_HIP_NSTAGES; ++i) {
        if (!set.recorded[i]) continue;
        if (prev >= 0) {
            HIP_TRY(hipEventSynchronize(set.ev[i]));
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, set.ev[prev], set.ev[i]));
            ms[prev] = t;
        }
        prev = i;
    }
    return ATSC_HIP_OK;
}

// stages a2..a4 + trellis parities (reference: atsc.cpp:80-82): everything that does not depend on the
// trellis state entering the batch
int run_front(atsc_hip_encoder* e, const uint8_t* d_ts, int fields, hipStream_t s) {
    const int C = e->cfg.channels;
    const size_t ts_stride = (size_t)fields * kFieldTs;
    mark(e, s, ATSC_HIP_STAGE_RAND_RS);
    launch_rand_rs(s, e->dtab, d_ts, ts_stride, e->buf, C, fields);
    if (e->buf.coded) launch_assemble(s, e->dtab, d_ts, ts_stride, e->buf, C, fields);
    mark(e, s, ATSC_HIP_STAGE_INTERLEAVE);
    launch_interleave(s, e->dtab, d_ts, ts_stride, e->buf, C, fields);  // + the per-part trellis parities
    std::swap(e->buf.tail, e->buf.tail_next);   // in stream order: the tail the launch wrote is now the history
    return static_cast<int>(hipGetLastError());
}

// carries + stage a5/a6 (reference: atsc.cpp:83-84)
int run_back(atsc_hip_encoder* e, const uint4* carry_xor, int fields, uint8_t* d_sym, size_t sym_stride, hipStream_t s) {
    const int C = e->cfg.channels;
    mark(e, s, ATSC_HIP_STAGE_TRELLIS_SCAN);
    launch_trellis_scan(s, e->buf, carry_xor, C, fields);
    mark(e, s, ATSC_HIP_STAGE_TRELLIS_MAP);
    launch_trellis_map(s, e->dtab, e->buf, d_sym, sym_stride, C, fields, e->field_index);
    return static_cast<int>(hipGetLastError());
}

// which form of the mixer + FIR this batch runs (include/atsc_hip.h, atsc_hip_set_fir_form).  A function of the handle's
// channel count and its fir_form setting only, so every rank of a time-split picks the same one for equal handles.
// The matrix-core kernels address a group of 16 channel rows through one buffer descriptor with 32-bit offsets: 16 rows of
// the largest batch's IQ (max_fields x 260416 samples) must stay below 4 GiB -- 128 fields of complex64, 257 of CS16.  A
// property of the handle (not of a batch), so that equal handles pick equal forms.
bool matrix_forms_fit(const atsc_hip_encoder* e, int iq_format, int upsample) {
    const uint64_t row_bytes = (uint64_t)e->cfg.max_fields * kFieldSyms * upsample * atsc_hip_iq_sample_bytes(iq_format);
    return 16 * row_bytes <= 0xffffffffull;
}

// The wide kernel (fir_w.hip) works on groups of 64 channels, writes complex64 at one sample per symbol, and addresses eight
// channel rows of the batch's IQ (and all 64 rows of the optional symbol tap) through 32-bit offsets.
bool wide_form_fits(const atsc_hip_encoder* e, int iq_format, int upsample) {
    if (iq_format != ATSC_HIP_IQ_CF32 || upsample != 1) return false;
    if (e->cfg.channels % kWideGroup) return false;   // whole groups only: its stores carry no range check
    const uint64_t row_bytes = (uint64_t)e->cfg.max_fields * kFieldSyms * sizeof(float2);
    return 8 * row_bytes <= 0xffffffffull && (uint64_t)kWideGroup * e->cfg.max_fields * kFieldSyms <= 0xffffffffull;
}

int resolve_fir_form(const atsc_hip_encoder* e) {
    if (e->fir_form != ATSC_HIP_FIR_AUTO) return e->fir_form;
    if (e->cfg.flags & ATSC_HIP_FLAG_STRICT_FP32) return ATSC_HIP_FIR_FFT;   // (set_upsample(2) is refused on such a handle)
    if (e->upsample != 1) return ATSC_HIP_FIR_FUSED;   // only the matrix-core kernel interpolates (set_upsample checked that it fits)
    if (!matrix_forms_fit(e, e->iq_format, 1)) return ATSC_HIP_FIR_FFT;
    // The fused kernel works on groups of kFusedGroup channels and costs the same for a partly filled group, the FFT
    // form costs per channel: fused when at least 80 % of the matrix columns carry a channel.
    // (the wide form is never AUTO's choice: at the bench shard it is 7 % slower than FUSED; it is the form to ask for when
    // the tighter error bound matters)
    const int groups = (e->cfg.channels + kFusedGroup - 1) / kFusedGroup;
    if (5 * e->cfg.channels >= 4 * kFusedGroup * groups) return ATSC_HIP_FIR_FUSED;
    return ATSC_HIP_FIR_FFT;
}

int run_mix_fir(atsc_hip_encoder* e, const uint8_t* d_sym, size_t sym_stride, void* d_iq, int fields, hipStream_t s) {
    const int C = e->cfg.channels;
    mark(e, s, ATSC_HIP_STAGE_OSC);
    const float2* oin = e->d_osc[e->osc_cur];
    float2* oout = e->d_osc[e->osc_cur ^ 1];
    if (d_iq) {
        const int form = resolve_fir_form(e);
        // (the fused form does not come through here, except for atsc_hip_mix_fir on a 2x handle: caller-supplied symbols
        // then take the matrix-core kernel's unfused instance, the only other one that interpolates)
        const bool mfma = form == ATSC_HIP_FIR_MFMA || e->upsample != 1;
        launch_osc_table(s, oin, oout, e->d_osc_tab, e->scale, fields, mfma ? kOscPlanes4096 : kOscComplex);
        mark(e, s, ATSC_HIP_STAGE_MIX_FIR);
        if (mfma)
            launch_mix_fir_mfma(s, d_sym, sym_stride, d_iq, (size_t)fields * kFieldSyms * e->upsample,
                                reinterpret_cast<const float*>(e->d_osc_tab), C, fields, e->iq_format, e->upsample);
        else
            launch_mix_fir_fft(s, d_sym, sym_stride, d_iq, (size_t)fields * kFieldSyms, e->d_osc_tab, C, fields, e->iq_format);
    } else {
        launch_osc_advance(s, oin, oout, e->scale, (unsigned)fields);
    }
    e->osc_cur ^= 1;
    return static_cast<int>(hipGetLastError());
}

}  // namespace

// The internal symbol buffer, on first use: only the separate-kernel forms called without a caller buffer write it (1 GiB at
// 64 channels x 64 fields that the fused form would hold for nothing).
static int ensure_sym_buffer(atsc_hip_encoder* e) {
    if (e->d_sym) return ATSC_HIP_OK;
    HIP_TRY(hipMalloc(&e->d_sym, (size_t)e->cfg.channels * e->sym_stride));
    return ATSC_HIP_OK;
}

static int check_batch_args(atsc_hip_encoder* e, const void* d_ts, int fields, const void* d_sym, const void* d_iq) {
    if (!e || !d_ts || fields < 1) return ATSC_HIP_ERR_ARG;
    if (fields > e->cfg.max_fields) return ATSC_HIP_ERR_TOO_LARGE;
    if ((reinterpret_cast<uintptr_t>(d_ts) & 3) || (reinterpret_cast<uintptr_t>(d_sym) & 15) ||
        (reinterpret_cast<uintptr_t>(d_iq) & 15))
        return ATSC_HIP_ERR_ARG;
    return ATSC_HIP_OK;
}

extern "C" int atsc_hip_encode_begin(atsc_hip_encoder* e, const uint8_t* d_ts, int fields, void* d_span_parity,
                                     void* stream) {
    int rc = check_batch_args(e, d_ts, fields, nullptr, d_span_parity);
    if (rc) return rc;
    if (e->pending_fields) return ATSC_HIP_ERR_STATE;
    HIP_TRY(hipSetDevice(e->cfg.device));
    auto s = static_cast<hipStream_t>(stream);
    begin_batch_timing(e);
    e->last_fields = fields;
    e->last_stream = s;
    rc = run_front(e, d_ts, fields, s);
    if (rc) return rc;
    if (d_span_parity) launch_span_parity(s, e->buf, static_cast<uint4*>(d_span_parity), e->cfg.channels, fields);
    e->pending_fields = fields;
    return static_cast<int>(hipGetLastError());
}

// After an error the batch is abandoned (pending_fields is cleared, so the handle accepts new batches), but the stream
// state (interleaver tail, carries, oscillator) may have been advanced part of the way: undefined until
// atsc_hip_import_state re-seats it (include/atsc_hip.h).
static int finish_impl(atsc_hip_encoder* e, const void* d_carry_xor, uint8_t* d_sym, void* d_iq, hipStream_t s, int fields) {
    e->last_fields = fields;
    e->last_stream = s;
    const size_t sym_stride = (size_t)fields * kFieldSyms;
    int rc;
    const int form = resolve_fir_form(e);
    const bool wide = d_iq && form == ATSC_HIP_FIR_WIDE;
    const bool fused = d_iq && (form == ATSC_HIP_FIR_FUSED || wide);
    if (!fused && !d_sym) {
        rc = ensure_sym_buffer(e);
        if (rc) return rc;
    }
    uint8_t* sym = d_sym ? d_sym : e->d_sym;
    bool fir_stop_recorded = false;
    if (fused) {
        // carries, then ONE kernel for trellis / map / sync / mixer / FIR: the symbols never pass through HBM; the symbol
        // tap is written only if the caller asked for it (d_sym)
        // (the carry scan, the oscillator table of the batch and the save of the batch's last interleaved bytes are one launch:
        // its time is reported under the oscillator stage, the scan and state stages read 0)
        mark(e, s, ATSC_HIP_STAGE_TRELLIS_SCAN);
        mark(e, s, ATSC_HIP_STAGE_OSC);
        launch_scan_osc_state(s, e->buf, static_cast<const uint4*>(d_carry_xor), e->cfg.channels, fields, e->d_osc[e->osc_cur],
                              e->d_osc[e->osc_cur ^ 1], e->d_osc_tab, e->scale, wide ? kOscPlanes16 : kOscPlanes4096);
        e->osc_cur ^= 1;
        // the mixer + FIR kernel's two events ride on its own dispatch when both are wanted (the usual case of a timed region)
        KernelEvents kev{};
        const bool own_events = !wide && e->cur && (e->mark_mask >> ATSC_HIP_STAGE_MIX_FIR & 1u) && (e->mark_mask >> ATSC_HIP_STAGE_STATE & 1u);
        if (own_events) {
            kev = {e->cur->ev[ATSC_HIP_STAGE_MIX_FIR], e->cur->ev[ATSC_HIP_STAGE_STATE]};
            e->cur->recorded[ATSC_HIP_STAGE_MIX_FIR] = e->cur->recorded[ATSC_HIP_STAGE_STATE] = true;
        } else {
            mark(e, s, ATSC_HIP_STAGE_MIX_FIR);
        }
        if (wide)
            launch_mix_fir_wide(s, e->dtab, e->buf, d_sym, sym_stride, d_iq, (size_t)fields * kFieldSyms,
                                reinterpret_cast<const float*>(e->d_osc_tab), e->cfg.channels, fields, e->field_index);
        else
        launch_mix_fir_fused(s, e->dtab, e->buf, d_sym, sym_stride, d_iq, (size_t)fields * kFieldSyms * e->upsample,
                             reinterpret_cast<const float*>(e->d_osc_tab), e->cfg.channels, fields, e->field_index,
                             e->iq_format, e->upsample, own_events ? &kev : nullptr);
        rc = static_cast<int>(hipGetLastError());
        if (rc) return rc;
        e->sym_internal = false;
        fir_stop_recorded = own_events;
    } else {
        rc = run_back(e, static_cast<const uint4*>(d_carry_xor), fields, sym, sym_stride, s);
        if (rc) return rc;
        e->sym_internal = d_sym == nullptr;
        rc = run_mix_fir(e, sym, sym_stride, d_iq, fields, s);
        if (rc) return rc;
    }
    if (!fir_stop_recorded) mark(e, s, ATSC_HIP_STAGE_STATE);
    if (!fused) launch_save_state(s, e->buf, e->cfg.channels, fields);   // (the fused path's merged launch has done it)
    std::swap(e->buf.last_group, e->buf.last_group_next);               // in stream order: what was saved is now the history
    mark(e, s, ATSC_HIP_NSTAGES);
    e->field_index += (uint64_t)fields;
    return static_cast<int>(hipGetLastError());
}

extern "C" int atsc_hip_encode_finish(atsc_hip_encoder* e, const void* d_carry_xor, uint8_t* d_sym, void* d_iq,
                                      void* stream) {
    if (!e) return ATSC_HIP_ERR_ARG;
    if (!e->pending_fields) return ATSC_HIP_ERR_STATE;
    if ((reinterpret_cast<uintptr_t>(d_sym) & 15) || (reinterpret_cast<uintptr_t>(d_iq) & 15) ||
        (reinterpret_cast<uintptr_t>(d_carry_xor) & 15))
        return ATSC_HIP_ERR_ARG;   // the batch stays pending: the caller may retry with valid pointers
    const int fields = e->pending_fields;
    e->pending_fields = 0;
    HIP_TRY(hipSetDevice(e->cfg.device));
    return finish_impl(e, d_carry_xor, d_sym, d_iq, static_cast<hipStream_t>(stream), fields);
}

extern "C" int atsc_hip_encode_batch(atsc_hip_encoder* e, const uint8_t* d_ts, int fields, uint8_t* d_sym, void* d_iq,
                                     void* stream) {
    int rc = check_batch_args(e, d_ts, fields, d_sym, d_iq);
    if (rc) return rc;
    rc = atsc_hip_encode_begin(e, d_ts, fields, nullptr, stream);
    if (rc) return rc;
    return atsc_hip_encode_finish(e, nullptr, d_sym, d_iq, stream);
}

extern "C" int atsc_hip_mix_fir(atsc_hip_encoder* e, const uint8_t* d_sym, int fields, void* d_iq, void* stream) {
    if (!e || !d_sym || !d_iq || fields < 1) return ATSC_HIP_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(d_sym) & 15) || (reinterpret_cast<uintptr_t>(d_iq) & 15)) return ATSC_HIP_ERR_ARG;
    if (fields > e->cfg.max_fields) return ATSC_HIP_ERR_TOO_LARGE;   // the batch's oscillator table holds max_fields fields
    if (e->pending_fields) return ATSC_HIP_ERR_STATE;                // would move the oscillator under the pending batch
    HIP_TRY(hipSetDevice(e->cfg.device));
    auto s = static_cast<hipStream_t>(stream);
    begin_batch_timing(e);
    e->last_stream = s;
    int rc = run_mix_fir(e, d_sym, (size_t)fields * kFieldSyms, d_iq, fields, s);
    mark(e, s, ATSC_HIP_STAGE_STATE);
    if (rc) return rc;
    e->field_index += (uint64_t)fields;
    return ATSC_HIP_OK;
}

extern "C" int atsc_hip_encode_batch_host(atsc_hip_encoder* e, const uint8_t* h_ts, int fields, uint8_t* h_sym,
                                          void* h_iq) {
    if (!e || !h_ts || fields < 1) return ATSC_HIP_ERR_ARG;
    if (fields > e->cfg.max_fields) return ATSC_HIP_ERR_TOO_LARGE;
    if (e->pending_fields) return ATSC_HIP_ERR_STATE;
    HIP_TRY(hipSetDevice(e->cfg.device));
    const size_t C = e->cfg.channels;
    const size_t ts_bytes = C * fields * kFieldTs, sym_bytes = C * fields * kFieldSyms;
    const size_t ss = atsc_hip_iq_sample_bytes(e->iq_format);
    uint8_t* d_ts = nullptr;
    uint8_t* d_sym = nullptr;   // a symbol tap of its own: the fused form writes symbols nowhere else
    void* d_iq = nullptr;
    int rc = static_cast<int>(hipMalloc(&d_ts, ts_bytes));
    if (!rc) rc = static_cast<int>(hipMemcpy(d_ts, h_ts, ts_bytes, hipMemcpyHostToDevice));
    if (!rc && h_sym) rc = static_cast<int>(hipMalloc(&d_sym, sym_bytes));
    if (!rc && h_iq) rc = static_cast<int>(hipMalloc(&d_iq, sym_bytes * ss * e->upsample));
    if (!rc) rc = atsc_hip_encode_batch(e, d_ts, fields, d_sym, d_iq, nullptr);
    if (!rc) rc = static_cast<int>(hipDeviceSynchronize());
    if (!rc && h_sym) rc = static_cast<int>(hipMemcpy(h_sym, d_sym, sym_bytes, hipMemcpyDeviceToHost));
    if (!rc && h_iq) rc = static_cast<int>(hipMemcpy(h_iq, d_iq, sym_bytes * ss * e->upsample, hipMemcpyDeviceToHost));
    hipFree(d_ts);
    hipFree(d_sym);
    hipFree(d_iq);
    return rc;
}

extern "C" size_t atsc_hip_iq_sample_bytes(int format) {
    switch (format) {
        case ATSC_HIP_IQ_CF32: return 8;
        case ATSC_HIP_IQ_CS16: return 4;
        case ATSC_HIP_IQ_CS8: return 2;
        default: return 0;
    }
}

extern "C" int atsc_hip_set_fir_form(atsc_hip_encoder* e, int form) {
    if (!e || form < ATSC_HIP_FIR_AUTO || form > ATSC_HIP_FIR_WIDE) return ATSC_HIP_ERR_ARG;
    if (e->pending_fields) return ATSC_HIP_ERR_STATE;
    if (form == ATSC_HIP_FIR_WIDE) {
        if (e->cfg.flags & ATSC_HIP_FLAG_STRICT_FP32) return ATSC_HIP_ERR_ARG;
        // complex64, one sample per symbol, whole groups of 64 channels only
        if (e->iq_format != ATSC_HIP_IQ_CF32 || e->upsample != 1 || e->cfg.channels % kWideGroup) return ATSC_HIP_ERR_ARG;
        if (!wide_form_fits(e, e->iq_format, e->upsample)) return ATSC_HIP_ERR_TOO_LARGE;
    }
    if ((form == ATSC_HIP_FIR_MFMA || form == ATSC_HIP_FIR_FUSED) && (e->cfg.flags & ATSC_HIP_FLAG_STRICT_FP32)) return ATSC_HIP_ERR_ARG;
    if ((form == ATSC_HIP_FIR_MFMA || form == ATSC_HIP_FIR_FUSED) && !matrix_forms_fit(e, e->iq_format, e->upsample)) return ATSC_HIP_ERR_TOO_LARGE;
    if (form == ATSC_HIP_FIR_FFT && e->upsample != 1) return ATSC_HIP_ERR_ARG;   // the FFT form does not interpolate
    e->fir_form = form;
    return ATSC_HIP_OK;
}

extern "C" int atsc_hip_get_fir_form(atsc_hip_encoder* e) {
    if (!e) return ATSC_HIP_ERR_ARG;
    return resolve_fir_form(e);
}

extern "C" int atsc_hip_set_iq_format(atsc_hip_encoder* e, int format) {
    if (!e || !atsc_hip_iq_sample_bytes(format)) return ATSC_HIP_ERR_ARG;
    if (e->pending_fields) return ATSC_HIP_ERR_STATE;
    if (e->fir_form == ATSC_HIP_FIR_WIDE && format != ATSC_HIP_IQ_CF32) return ATSC_HIP_ERR_ARG;
    if ((e->fir_form == ATSC_HIP_FIR_MFMA || e->fir_form == ATSC_HIP_FIR_FUSED || e->upsample != 1) && !matrix_forms_fit(e, format, e->upsample))
        return ATSC_HIP_ERR_TOO_LARGE;
    e->iq_format = format;
    return ATSC_HIP_OK;
}

extern "C" int atsc_hip_set_upsample(atsc_hip_encoder* e, int factor) {
    if (!e || (factor != 1 && factor != 2)) return ATSC_HIP_ERR_ARG;
    if (e->pending_fields) return ATSC_HIP_ERR_STATE;
    if (factor == e->upsample) return ATSC_HIP_OK;
    if (factor == 2) {
        if (e->fir_form == ATSC_HIP_FIR_WIDE) return ATSC_HIP_ERR_ARG;
        if (e->fir_form == ATSC_HIP_FIR_FFT || (e->cfg.flags & ATSC_HIP_FLAG_STRICT_FP32)) return ATSC_HIP_ERR_ARG;
        if (!matrix_forms_fit(e, e->iq_format, 2)) return ATSC_HIP_ERR_TOO_LARGE;
    }
    e->upsample = factor;
    // the streaming slots are sized for the output rate: released here, allocated again by the next atsc_hip_process call
    for (auto& sl : e->slot) {
        hipFree(sl.d_iq);
        sl.d_iq = nullptr;
        if (sl.h_iq) hipHostFree(sl.h_iq);
        sl.h_iq = nullptr;
    }
    return ATSC_HIP_OK;
}

extern "C" int atsc_hip_get_upsample(atsc_hip_encoder* e) { return e ? e->upsample : ATSC_HIP_ERR_ARG; }

// Streaming path: reference semantics of atsc_encoder_impl::process (atsc.cpp:59-77).  With mtu_samples > 0 every
// field is handed out in pieces of at most mtu_samples, never spanning a field (util/atsc_encoder.cpp:90-99).
//
// A call that completes several batches runs them as a two-slot pipeline: upload + kernels of batch k+1 (compute
// stream) overlap the download of batch k (copy stream) and its callbacks (this thread).  The callbacks stay
// synchronous, in field order, on the caller's thread, and every field completed by this call has been delivered
// when the call returns (atsc.cpp:72-75,89) -- nothing is left in flight between calls.
namespace {

int streaming_setup(atsc_hip_encoder* e) {
    const size_t F = e->cfg.max_fields;
    if (!e->compute_stream) HIP_TRY(hipStreamCreateWithFlags(&e->compute_stream, hipStreamNonBlocking));
    if (!e->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    // sized for CF32 (the largest sample format); each piece is retried if an earlier call ran out of memory
    for (auto& sl : e->slot) {
        if (!sl.d_ts) HIP_TRY(hipMalloc(&sl.d_ts, F * kFieldTs));
        if (!sl.d_iq) HIP_TRY(hipMalloc(&sl.d_iq, F * kFieldSyms * e->upsample * sizeof(float2)));
        if (!sl.h_ts) HIP_TRY(hipHostMalloc(&sl.h_ts, F * kFieldTs, hipHostMallocDefault));
        if (!sl.h_iq) HIP_TRY(hipHostMalloc(&sl.h_iq, F * kFieldSyms * e->upsample * sizeof(float2), hipHostMallocDefault));
        if (!sl.encoded) HIP_TRY(hipEventCreateWithFlags(&sl.encoded, hipEventDisableTiming));
        if (!sl.copied) HIP_TRY(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
    }
    return ATSC_HIP_OK;
}

// Bytes of e->pending that were handed to the GPU are dropped on EVERY way out of process_impl (normal return, HIP
// error, a callback that throws): those fields have advanced the stream state and must not be encoded again.
struct ConsumedGuard {
    atsc_hip_encoder* e;
    size_t done = 0;
    ~ConsumedGuard() {
        if (e->compute_stream) hipStreamSynchronize(e->compute_stream);   // nothing stays in flight between calls
        if (e->copy_stream) hipStreamSynchronize(e->copy_stream);
        if (done) e->pending.erase(e->pending.begin(), e->pending.begin() + done);
    }
};

}  // namespace

static int process_impl(atsc_hip_encoder* e, const uint8_t* h_pkts, unsigned count, atsc_hip_field_cb cb,
                        atsc_hip_chunk_cb chunk_cb, size_t mtu_samples, void* user) {
    if (!e || (!h_pkts && count)) return ATSC_HIP_ERR_ARG;
    if (e->cfg.channels != 1 || e->pending_fields) return ATSC_HIP_ERR_STATE;
    for (const auto& q : e->multi.fifo)
        if (!q.empty()) return ATSC_HIP_ERR_STATE;   // the stream is being fed through atsc_hip_process_multi
    HIP_TRY(hipSetDevice(e->cfg.device));
    int rc = streaming_setup(e);
    if (rc) return rc;
    // The streaming path runs on private non-blocking streams: whatever the caller queued for this handle on another
    // stream (atsc_hip_encode_batch, atsc_hip_seek) must have touched the tail / carry / oscillator buffers first
    if (e->last_stream != e->compute_stream) {
        if (!e->order_ev) HIP_TRY(hipEventCreateWithFlags(&e->order_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(e->order_ev, e->last_stream));
        HIP_TRY(hipStreamWaitEvent(e->compute_stream, e->order_ev, 0));
    }
    const size_t F = e->cfg.max_fields;
    const size_t ss = atsc_hip_iq_sample_bytes(e->iq_format);
    const size_t field_samples = (size_t)kFieldSyms * e->upsample;
    const size_t field_bytes = field_samples * ss;
    // copy-before-return contract (atsc.cpp:66); no exception may cross the C boundary
    try {
        e->pending.insert(e->pending.end(), h_pkts, h_pkts + (size_t)count * ATSC_HIP_TS_PACKET_BYTES);
    } catch (const std::bad_alloc&) {
        return static_cast<int>(hipErrorOutOfMemory);
    }
    ConsumedGuard consumed{e};
    int nf_of[2] = {0, 0};

    // upload + kernels + download of the next batch, into slot k & 1; 0 fields if no whole field is left
    auto submit = [&](unsigned k) -> int {
        auto& sl = e->slot[k & 1];
        const int nf = (int)std::min<size_t>((e->pending.size() - consumed.done) / kFieldTs, F);
        nf_of[k & 1] = nf;
        if (nf == 0) return ATSC_HIP_OK;
        std::memcpy(sl.h_ts, e->pending.data() + consumed.done, (size_t)nf * kFieldTs);
        HIP_TRY(hipMemcpyAsync(sl.d_ts, sl.h_ts, (size_t)nf * kFieldTs, hipMemcpyHostToDevice, e->compute_stream));
        consumed.done += (size_t)nf * kFieldTs;   // from here on the stream state moves: the bytes count as consumed
        const int r = atsc_hip_encode_batch(e, sl.d_ts, nf, nullptr, sl.d_iq, e->compute_stream);
        if (r) return r;
        HIP_TRY(hipEventRecord(sl.encoded, e->compute_stream));
        HIP_TRY(hipStreamWaitEvent(e->copy_stream, sl.encoded, 0));
        HIP_TRY(hipMemcpyAsync(sl.h_iq, sl.d_iq, (size_t)nf * field_bytes, hipMemcpyDeviceToHost, e->copy_stream));
        HIP_TRY(hipEventRecord(sl.copied, e->copy_stream));
        return ATSC_HIP_OK;
    };

    rc = submit(0);
    if (rc) return rc;
    for (unsigned k = 0; nf_of[k & 1] > 0; ++k) {
        // slot (k+1) & 1 is free: its previous batch (k-1) was delivered in the last trip
        rc = submit(k + 1);
        if (rc) return rc;
        auto& sl = e->slot[k & 1];
        HIP_TRY(hipEventSynchronize(sl.copied));
        for (int f = 0; f < nf_of[k & 1]; ++f) {
            uint8_t* field = sl.h_iq + (size_t)f * field_bytes;
            if (cb) cb(user, field, (unsigned)field_bytes);
            if (chunk_cb) {
                size_t remaining = field_samples;
                while (remaining > 0) {
                    const size_t n = std::min(remaining, mtu_samples);
                    chunk_cb(user, field + (field_samples - remaining) * ss, n, e->iq_format);
                    remaining -= n;
                }
            }
        }
    }
    return ATSC_HIP_OK;
}

extern "C" int atsc_hip_process(atsc_hip_encoder* e, const uint8_t* h_pkts, unsigned count, atsc_hip_field_cb cb,
                                void* user) {
    return process_impl(e, h_pkts, count, cb, nullptr, 0, user);
}

extern "C" int atsc_hip_process_chunked(atsc_hip_encoder* e, const uint8_t* h_pkts, unsigned count, size_t mtu_samples,
                                        atsc_hip_chunk_cb cb, void* user) {
    if (!mtu_samples || !cb) return ATSC_HIP_ERR_ARG;
    return process_impl(e, h_pkts, count, nullptr, cb, mtu_samples, user);
}

// ---- multi-stream front: K independent TS streams on ONE K-channel handle (reference: one atsc_encoder per stream,
// src/atsc/atsc.h:7-16, atsc.cpp:59-77; util/atsc_encoder.cpp:83-100).  Every stream keeps process()'s contract -- any packet
// count, bytes copied before return, one callback per completed field, in order, on a calling thread -- but the fields of
// all streams go through the GPU together, as batches of the many-channel kernels (the fused matrix-core form when the
// channel count fills its groups), which is where the throughput of this library is.
namespace {

int multi_setup(atsc_hip_encoder* e) {
    auto& m = e->multi;
    const size_t K = e->cfg.channels, F = e->cfg.max_fields;
    try {
        if (m.fifo.size() != K) {
            m.fifo.assign(K, {});
            m.cb.assign(K, nullptr);
            m.user.assign(K, nullptr);
        }
    } catch (const std::bad_alloc&) {
        return static_cast<int>(hipErrorOutOfMemory);
    }
    if (!e->compute_stream) HIP_TRY(hipStreamCreateWithFlags(&e->compute_stream, hipStreamNonBlocking));
    if (!e->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    // the IQ buffers follow the handle's current sample format and rate; they are rebuilt when a switch needs larger ones
    const size_t iq_bytes = K * F * kFieldSyms * e->upsample * atsc_hip_iq_sample_bytes(e->iq_format);
    for (auto& sl : m.slot) {
        if (iq_bytes > m.slot_iq_bytes) {
            if (sl.d_iq) { HIP_TRY(hipFree(sl.d_iq)); sl.d_iq = nullptr; }
            if (sl.h_iq) { HIP_TRY(hipHostFree(sl.h_iq)); sl.h_iq = nullptr; }
        }
        if (!sl.d_ts) HIP_TRY(hipMalloc(&sl.d_ts, K * F * kFieldTs));
        if (!sl.h_ts) HIP_TRY(hipHostMalloc(&sl.h_ts, K * F * kFieldTs, hipHostMallocDefault));
        if (!sl.d_iq) HIP_TRY(hipMalloc(&sl.d_iq, iq_bytes));
        if (!sl.h_iq) HIP_TRY(hipHostMalloc(&sl.h_iq, iq_bytes, hipHostMallocDefault));
        if (!sl.encoded) HIP_TRY(hipEventCreateWithFlags(&sl.encoded, hipEventDisableTiming));
        if (!sl.copied) HIP_TRY(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
    }
    if (iq_bytes > m.slot_iq_bytes) m.slot_iq_bytes = iq_bytes;
    return ATSC_HIP_OK;
}

// Bytes handed to the GPU leave the FIFOs on every way out (normal return, HIP error, a callback that throws): those
// fields have advanced the streams' state and must not be encoded again.
struct MultiConsumedGuard {
    atsc_hip_encoder* e;
    size_t done = 0;   // bytes per channel
    ~MultiConsumedGuard() {
        if (e->compute_stream) hipStreamSynchronize(e->compute_stream);
        if (e->copy_stream) hipStreamSynchronize(e->copy_stream);
        if (done)
            for (auto& q : e->multi.fifo) q.erase(q.begin(), q.begin() + done);
    }
};

}  // namespace

extern "C" int atsc_hip_process_multi(atsc_hip_encoder* e, int channel, const uint8_t* h_pkts, unsigned count,
                                      atsc_hip_field_cb cb, void* user) {
    if (!e || channel < 0 || channel >= e->cfg.channels || (!h_pkts && count)) return ATSC_HIP_ERR_ARG;
    if (e->pending_fields || !e->pending.empty()) return ATSC_HIP_ERR_STATE;   // a batch in flight, or atsc_hip_process data buffered
    HIP_TRY(hipSetDevice(e->cfg.device));
    int rc = multi_setup(e);
    if (rc) return rc;
    auto& m = e->multi;
    const size_t K = e->cfg.channels, F = e->cfg.max_fields;
    auto& mine = m.fifo[channel];
    // a stream may run ahead of the slowest one by ATSC_HIP_MULTI_MAX_AHEAD batches, no further: the FIFOs are host memory
    if (mine.size() + (size_t)count * ATSC_HIP_TS_PACKET_BYTES > ((size_t)ATSC_HIP_MULTI_MAX_AHEAD * F + 1) * kFieldTs)
        return ATSC_HIP_ERR_TOO_LARGE;
    try {
        mine.insert(mine.end(), h_pkts, h_pkts + (size_t)count * ATSC_HIP_TS_PACKET_BYTES);   // copy-before-return (atsc.cpp:66)
    } catch (const std::bad_alloc&) {
        return static_cast<int>(hipErrorOutOfMemory);
    }
    m.cb[channel] = cb;
    m.user[channel] = user;
    size_t ready = SIZE_MAX;   // whole fields every stream has
    for (const auto& q : m.fifo) ready = std::min(ready, q.size() / kFieldTs);
    if (ready == 0) return ATSC_HIP_OK;

    if (e->last_stream != e->compute_stream) {   // as in atsc_hip_process: behind whatever touched the handle's state last
        if (!e->order_ev) HIP_TRY(hipEventCreateWithFlags(&e->order_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(e->order_ev, e->last_stream));
        HIP_TRY(hipStreamWaitEvent(e->compute_stream, e->order_ev, 0));
    }
    const size_t ss = atsc_hip_iq_sample_bytes(e->iq_format);
    const size_t field_bytes = (size_t)kFieldSyms * e->upsample * ss;
    MultiConsumedGuard consumed{e};
    int nf_of[2] = {0, 0};
    auto submit = [&](unsigned k) -> int {
        auto& sl = m.slot[k & 1];
        const int nf = (int)std::min<size_t>(ready - consumed.done / kFieldTs, F);
        nf_of[k & 1] = nf;
        if (nf == 0) return ATSC_HIP_OK;
        for (size_t c = 0; c < K; ++c)   // [channel][field][312 x 188], packed with this batch's field count
            std::memcpy(sl.h_ts + c * nf * kFieldTs, m.fifo[c].data() + consumed.done, (size_t)nf * kFieldTs);
        HIP_TRY(hipMemcpyAsync(sl.d_ts, sl.h_ts, K * nf * kFieldTs, hipMemcpyHostToDevice, e->compute_stream));
        consumed.done += (size_t)nf * kFieldTs;
        const int r = atsc_hip_encode_batch(e, sl.d_ts, nf, nullptr, sl.d_iq, e->compute_stream);
        if (r) return r;
        HIP_TRY(hipEventRecord(sl.encoded, e->compute_stream));
        HIP_TRY(hipStreamWaitEvent(e->copy_stream, sl.encoded, 0));
        HIP_TRY(hipMemcpyAsync(sl.h_iq, sl.d_iq, K * nf * field_bytes, hipMemcpyDeviceToHost, e->copy_stream));
        HIP_TRY(hipEventRecord(sl.copied, e->copy_stream));
        return ATSC_HIP_OK;
    };
    rc = submit(0);
    if (rc) return rc;
    for (unsigned k = 0; nf_of[k & 1] > 0; ++k) {
        rc = submit(k + 1);   // upload + kernels of the next batch overlap this one's download and callbacks
        if (rc) return rc;
        auto& sl = m.slot[k & 1];
        HIP_TRY(hipEventSynchronize(sl.copied));
        const int nf = nf_of[k & 1];
        for (int f = 0; f < nf; ++f)        // field-major: the streams move on together
            for (size_t c = 0; c < K; ++c)
                if (m.cb[c]) m.cb[c](m.user[c], sl.h_iq + (c * nf + f) * field_bytes, (unsigned)field_bytes);
    }
    return ATSC_HIP_OK;
}

extern "C" int atsc_hip_multi_pending(atsc_hip_encoder* e, int channel, size_t* packets) {
    if (!e || !packets || channel < 0 || channel >= e->cfg.channels) return ATSC_HIP_ERR_ARG;
    *packets = (size_t)channel < e->multi.fifo.size() ? e->multi.fifo[channel].size() / ATSC_HIP_TS_PACKET_BYTES : 0;
    return ATSC_HIP_OK;
}

extern "C" int atsc_hip_export_state(atsc_hip_encoder* e, int channel, atsc_hip_stream_state* out) {
    if (!e || !out || channel < 0 || channel >= e->cfg.channels) return ATSC_HIP_ERR_ARG;
    HIP_TRY(hipSetDevice(e->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    std::memset(out, 0, sizeof *out);
    out->field_index = e->field_index;
    HIP_TRY(hipMemcpy(out->coded_tail, e->buf.tail + (size_t)channel * kTail, kTail, hipMemcpyDeviceToHost));
    uint8_t lg[16];
    HIP_TRY(hipMemcpy(lg, e->buf.last_group + (size_t)channel * 16, 16, hipMemcpyDeviceToHost));
    std::memcpy(out->last_group, lg, 12);
    uint32_t cw[4];
    HIP_TRY(hipMemcpy(cw, e->buf.carry + channel, 16, hipMemcpyDeviceToHost));
    for (int en = 0; en < 12; ++en) {
        const uint32_t b = (cw[en / 4] >> (8 * (en % 4))) & 0xff;
        out->z2[en] = b & 1;
        out->z0_next[en] = (b >> 1) & 1;
        out->z0_prev[en] = (b >> 2) & 1;
    }
    return ATSC_HIP_OK;
}

extern "C" int atsc_hip_import_state(atsc_hip_encoder* e, int channel, const atsc_hip_stream_state* in) {
    if (!e || !in || channel < 0 || channel >= e->cfg.channels) return ATSC_HIP_ERR_ARG;
    HIP_TRY(hipSetDevice(e->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(e->buf.tail + (size_t)channel * kTail, in->coded_tail, kTail, hipMemcpyHostToDevice));
    uint8_t lg[16] = {0};
    std::memcpy(lg, in->last_group, 12);
    HIP_TRY(hipMemcpy(e->buf.last_group + (size_t)channel * 16, lg, 16, hipMemcpyHostToDevice));
    uint32_t cw[4] = {0, 0, 0, 0};
    for (int en = 0; en < 12; ++en) {
        const uint32_t b = (in->z2[en] & 1u) | ((in->z0_next[en] & 1u) << 1) | ((in->z0_prev[en] & 1u) << 2);
        cw[en / 4] |= b << (8 * (en % 4));
    }
    HIP_TRY(hipMemcpy(e->buf.carry + channel, cw, 16, hipMemcpyHostToDevice));
    if (in->field_index != e->field_index) {
        if (in->field_index > ATSC_HIP_MAX_SEEK_STEPS) return ATSC_HIP_ERR_TOO_LARGE;
        // T_f = T_0 (x) scale^f by the reference's own sequential fp32 recursion (offset.h:35-41)
        HIP_TRY(hipMemcpy(e->d_osc[e->osc_cur ^ 1], e->tab.osc0.data(), sizeof(float2) * kPaddedSyms, hipMemcpyHostToDevice));
        uint64_t left = in->field_index;
        int src = e->osc_cur ^ 1;
        while (left > 0) {
            const unsigned n = (unsigned)std::min<uint64_t>(left, 1u << 20);
            launch_osc_advance(nullptr, e->d_osc[src], e->d_osc[src ^ 1], e->scale, n);
            src ^= 1;
            left -= n;
        }
        HIP_TRY(hipDeviceSynchronize());
        e->osc_cur = src;
        e->field_index = in->field_index;
    }
    return ATSC_HIP_OK;
}

extern "C" int atsc_hip_seek(atsc_hip_encoder* e, uint64_t field_index, void* stream) {
    if (!e) return ATSC_HIP_ERR_ARG;
    if (e->pending_fields) return ATSC_HIP_ERR_STATE;
    // T_f = T_(f-1) (x) scale, sequentially in fp32 (offset.h:35-41): the handle's state IS T_(its field index) of that
    // sequence, so a target ahead of it is reached from there; one behind it from the device copy of T_0.  The walk is
    // one dependent step per field and lane: more than ATSC_HIP_MAX_SEEK_STEPS of them are refused (a garbage index would
    // otherwise hang the stream for minutes)
    const bool ahead = field_index >= e->field_index;
    const uint64_t steps = ahead ? field_index - e->field_index : field_index;
    if (steps > ATSC_HIP_MAX_SEEK_STEPS) return ATSC_HIP_ERR_TOO_LARGE;
    HIP_TRY(hipSetDevice(e->cfg.device));
    auto s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(e->buf.carry, 0, sizeof(uint4) * e->cfg.channels, s));
    e->last_stream = s;
    if (field_index != e->field_index) {
        const float2* from = ahead ? e->d_osc[e->osc_cur] : e->d_osc0;
        uint64_t left = steps;
        int dst = e->osc_cur ^ 1;
        if (left == 0) {
            HIP_TRY(hipMemcpyAsync(e->d_osc[dst], from, sizeof(float2) * kPaddedSyms, hipMemcpyDeviceToDevice, s));
        } else {
            while (left > 0) {
                const unsigned n = (unsigned)std::min<uint64_t>(left, 1u << 20);
                launch_osc_advance(s, from, e->d_osc[dst], e->scale, n);
                from = e->d_osc[dst];
                left -= n;
                if (left) dst ^= 1;
            }
        }
        HIP_TRY(hipGetLastError());   // host state moves only when every launch was accepted
        e->osc_cur = dst;
        e->field_index = field_index;
    }
    return ATSC_HIP_OK;
}

extern "C" int atsc_hip_span_parity(atsc_hip_encoder* e, int channel, uint8_t* out36) {
    if (!e || !out36 || channel < 0 || channel >= e->cfg.channels) return ATSC_HIP_ERR_ARG;
    HIP_TRY(hipSetDevice(e->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    uint32_t cw[4];
    HIP_TRY(hipMemcpy(cw, e->buf.span_par + channel, 16, hipMemcpyDeviceToHost));
    for (int en = 0; en < 12; ++en) {
        const uint32_t b = (cw[en / 4] >> (8 * (en % 4))) & 0xff;
        out36[en] = b & 1;
        out36[12 + en] = (b >> 1) & 1;
        out36[24 + en] = (b >> 2) & 1;
    }
    return ATSC_HIP_OK;
}

extern "C" int atsc_hip_stage_buffer(atsc_hip_encoder* e, int stage, void** d_ptr, size_t* bytes, size_t* channel_stride) {
    if (!e || !d_ptr) return ATSC_HIP_ERR_ARG;
    size_t b = 0, st = 0;
    void* p = nullptr;
    switch (stage) {
        case ATSC_HIP_STAGE_RAND_RS:
            if (!e->buf.coded) return ATSC_HIP_ERR_STATE;  // handle created without ATSC_HIP_FLAG_TAPS
            p = e->buf.coded; st = e->buf.coded_stride; b = st * e->cfg.channels; break;
        case ATSC_HIP_STAGE_INTERLEAVE: p = e->buf.il; st = e->buf.il_stride; b = st * e->cfg.channels; break;
        // par / cin are packed with the batch's own field count: [channel][last_fields * 26] x 16 B
        case ATSC_HIP_STAGE_TRELLIS_PAR: p = e->buf.par; st = (size_t)e->last_fields * kParts * 16; b = st * e->cfg.channels; break;
        case ATSC_HIP_STAGE_TRELLIS_SCAN: p = e->buf.cin; st = (size_t)e->last_fields * kParts * 16; b = st * e->cfg.channels; break;
        case ATSC_HIP_STAGE_TRELLIS_MAP:
            if (!e->sym_internal) return ATSC_HIP_ERR_STATE;  // symbols went to the caller's buffer, or (fused form) nowhere
            p = e->d_sym; st = (size_t)e->last_fields * kFieldSyms; b = st * e->cfg.channels; break;
        case ATSC_HIP_STAGE_MIX_FIR: p = e->d_osc[e->osc_cur]; st = 0; b = sizeof(float2) * kPaddedSyms; break;
        default: return ATSC_HIP_ERR_ARG;
    }
    *d_ptr = p;
    if (bytes) *bytes = b;
    if (channel_stride) *channel_stride = st;
    return ATSC_HIP_OK;
}

extern "C" int atsc_hip_last_timings(atsc_hip_encoder* e, float* ms) {
    if (!e || !ms) return ATSC_HIP_ERR_ARG;
    if (!e->ev_valid || !e->cur) return ATSC_HIP_ERR_STATE;
    HIP_TRY(hipSetDevice(e->cfg.device));
    return read_timings(e, *e->cur, ms);
}

extern "C" int atsc_hip_timing_reserve(atsc_hip_encoder* e, int nslots) {
    if (!e || nslots < 1) return ATSC_HIP_ERR_ARG;
    if (!e->ev_valid) return ATSC_HIP_ERR_STATE;
    HIP_TRY(hipSetDevice(e->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t old = e->slots.size();
    if ((size_t)nslots > old) {
        e->slots.resize(nslots);
        for (size_t i = old; i < e->slots.size(); ++i)
            for (auto& ev : e->slots[i].ev) HIP_TRY(hipEventCreate(&ev));
    }
    for (auto& set : e->slots) std::fill(std::begin(set.recorded), std::end(set.recorded), false);
    e->batches = 0;
    e->cur = nullptr;
    return ATSC_HIP_OK;
}

extern "C" int atsc_hip_timing_select(atsc_hip_encoder* e, unsigned stage_mask) {
    if (!e) return ATSC_HIP_ERR_ARG;
    if (!e->ev_valid) return ATSC_HIP_ERR_STATE;
    // a stage's time runs from its own mark to the next recorded one: the mark behind a selected stage is recorded as well
    unsigned m = 0;
    for (int i = 0; i < ATSC_HIP_NSTAGES; ++i)
        if (stage_mask >> i & 1u) m |= 3u << i;
    e->mark_mask = stage_mask == ~0u ? ~0u : m;
    return ATSC_HIP_OK;
}

extern "C" int atsc_hip_batch_timings(atsc_hip_encoder* e, int batch, float* ms) {
    if (!e || !ms || batch < 0) return ATSC_HIP_ERR_ARG;
    if (!e->ev_valid) return ATSC_HIP_ERR_STATE;
    if ((size_t)batch >= e->batches || e->batches - (size_t)batch > e->slots.size()) return ATSC_HIP_ERR_ARG;
    HIP_TRY(hipSetDevice(e->cfg.device));
    return read_timings(e, e->slots[(size_t)batch % e->slots.size()], ms);
}

extern "C" int atsc_hip_get_table(atsc_hip_encoder* e, int which, void* h_out, size_t bytes) {
    if (!h_out) return ATSC_HIP_ERR_ARG;
    // enc == NULL: build the (device-independent) tables on the fly; needs no GPU.
    atsc_tables::Tables local;
    const atsc_tables::Tables* t = e ? &e->tab : &local;
    if (!e) atsc_tables::build(local);
    const void* src = nullptr;
    size_t n = 0;
    switch (which) {
        case 0: src = t->taps; n = sizeof(float) * kTaps; break;
        case 1: src = t->osc0.data(); n = sizeof(float) * 2 * kPaddedSyms; break;
        case 2: src = t->scale; n = sizeof(float) * 2; break;
        case 3: src = t->prbs.data(); n = t->prbs.size(); break;
        case 4: src = t->fsync[0].data(); n = 820; break;
        case 5: src = t->fsync[1].data(); n = 820; break;
        case 6: src = t->rs_gen; n = 21; break;
        case 7: src = t->taps_half; n = sizeof(float) * kTaps; break;
        case 8:
        case 9: {
            // the matrix-core forms' tap operands as fp16 bit patterns: with a handle, read back from its device (what the
            // kernels use); without, built on the host by the function whose output create() uploads
            n = sizeof(uint16_t) * (which == 8 ? kMmOperandHalves : kWOperandHalves);
            if (bytes < n) return ATSC_HIP_ERR_ARG;
            if (e) {
                HIP_TRY(hipSetDevice(e->cfg.device));
                HIP_TRY(which == 8 ? download_mm_tables(static_cast<uint16_t*>(h_out)) : download_w_tables(static_cast<uint16_t*>(h_out)));
                return ATSC_HIP_OK;
            }
            std::vector<uint16_t> op;
            if (which == 8) build_mm_operand(t->taps, t->taps_half, op);
            else build_w_operand(t->taps, op);
            std::memcpy(h_out, op.data(), n);
            return ATSC_HIP_OK;
        }
        default: return ATSC_HIP_ERR_ARG;
    }
    if (bytes < n) return ATSC_HIP_ERR_ARG;
    std::memcpy(h_out, src, n);
    return ATSC_HIP_OK;
}
