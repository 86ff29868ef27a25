"""GPU tests added in round 3: the host-layer items of VERDICT round 2 (strict-fp32 handles, bounded seeks, the lazily
allocated symbol buffer, stream order between the batch and the streaming calls of one handle).  Everything goes through
the C-ABI."""
import ctypes as C

import numpy as np
import pytest

import atsc_amd
from atsc_amd import FIELD_SYMBOLS, FIELD_TS_BYTES
from oracle_lib import Oracle, ts_mt
from test_gpu_parity import IQ_TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _code(fn, *a):
    with pytest.raises(atsc_amd.AtscHipError) as ei:
        fn(*a)
    return ei.value.code


def test_strict_fp32_handle_never_runs_a_matrix_core_form(torch_cuda):
    """ATSC_HIP_FLAG_STRICT_FP32: AUTO resolves to the fp32 (FFT) form whatever the channel count, the matrix-core forms
    and the 2x form are refused; a 16-channel strict handle and a 1-channel handle give the same IQ bits for the same
    stream (the bits no longer depend on the grouping)."""
    torch = torch_cuda
    enc = atsc_amd.Encoder(channels=16, max_fields=2, strict_fp32=True)
    assert enc.fir_form() == atsc_amd.FIR_FFT
    assert atsc_amd.Encoder(channels=16, max_fields=2).fir_form() == atsc_amd.FIR_FUSED
    assert _code(enc.set_fir_form, atsc_amd.FIR_FUSED) == atsc_amd.ERR_ARG
    assert _code(enc.set_fir_form, atsc_amd.FIR_MFMA) == atsc_amd.ERR_ARG
    assert _code(enc.set_upsample, 2) == atsc_amd.ERR_ARG
    enc.set_fir_form(atsc_amd.FIR_FFT)
    enc.set_fir_form(atsc_amd.FIR_AUTO)
    nf = 2
    ts = torch.from_numpy(np.stack([ts_mt(300 + c, nf) for c in range(16)])).cuda()
    iq16 = torch.empty(16 * nf * FIELD_SYMBOLS, dtype=torch.complex64, device="cuda")
    enc.encode_batch(ts.data_ptr(), nf, None, iq16.data_ptr(), torch.cuda.current_stream().cuda_stream)
    one = atsc_amd.Encoder(channels=1, max_fields=nf)
    iq1 = torch.empty(nf * FIELD_SYMBOLS, dtype=torch.complex64, device="cuda")
    one.encode_batch(ts[5].contiguous().data_ptr(), nf, None, iq1.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(torch.view_as_real(iq16.view(16, -1)[5]), torch.view_as_real(iq1))


def test_seek_refuses_walks_beyond_the_bound(torch_cuda):
    """atsc_hip_seek / atsc_hip_import_state: more than ATSC_HIP_MAX_SEEK_STEPS sequential oscillator steps are refused
    with ATSC_HIP_ERR_TOO_LARGE and leave the handle where it was."""
    torch = torch_cuda
    enc = atsc_amd.Encoder(channels=1, max_fields=1)
    assert _code(enc.seek, atsc_amd.MAX_SEEK_STEPS + 1, None) == atsc_amd.ERR_TOO_LARGE
    assert _code(enc.seek, (1 << 63) + 12345, None) == atsc_amd.ERR_TOO_LARGE
    st = enc.export_state(0)
    assert st.field_index == 0
    st.field_index = atsc_amd.MAX_SEEK_STEPS + 7
    assert _code(enc.import_state, st, 0) == atsc_amd.ERR_TOO_LARGE
    enc.seek(3, None)
    # a later seek counts its steps from the handle's own index when the target lies ahead
    enc.seek(5, None)
    torch.cuda.synchronize()
    assert enc.export_state(0).field_index == 5
    ref = atsc_amd.Encoder(channels=1, max_fields=1)
    ref.seek(5, None)
    a, _, _ = enc.stage_buffer(atsc_amd.STAGE_MIX_FIR)
    b, n, _ = ref.stage_buffer(atsc_amd.STAGE_MIX_FIR)
    import ctypes as C
    ta = torch.empty(n, dtype=torch.uint8, device="cuda")
    tb = torch.empty(n, dtype=torch.uint8, device="cuda")
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(ta.data_ptr(), a, n, 3) == 0 and hip.hipMemcpy(tb.data_ptr(), b, n, 3) == 0
    assert torch.equal(ta, tb), "T_5 reached as 0 -> 3 -> 5 differs from T_5 reached in one walk"


def test_internal_symbol_buffer_is_allocated_on_first_need(torch_cuda):
    """A handle that runs the fused form holds no [channels][fields][260416] symbol buffer (1 GiB at the bench shard);
    the separate-kernel forms allocate it with their first batch that has no caller buffer."""
    torch = torch_cuda
    C_, nf = 32, 8
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    enc = atsc_amd.Encoder(channels=C_, max_fields=nf)
    sym_bytes = C_ * nf * FIELD_SYMBOLS
    held = free0 - torch.cuda.mem_get_info()[0]
    ts = torch.zeros(C_ * nf * FIELD_TS_BYTES, dtype=torch.uint8, device="cuda")
    iq = torch.empty(C_ * nf * FIELD_SYMBOLS, dtype=torch.complex64, device="cuda")
    free1 = torch.cuda.mem_get_info()[0]
    enc.encode_batch(ts.data_ptr(), nf, None, iq.data_ptr(), None)
    torch.cuda.synchronize()
    assert enc.fir_form() == atsc_amd.FIR_FUSED
    assert free1 - torch.cuda.mem_get_info()[0] < sym_bytes // 2, "the fused batch allocated a symbol buffer"
    assert _code(enc.stage_buffer, atsc_amd.STAGE_TRELLIS_MAP) == atsc_amd.ERR_STATE
    enc.set_fir_form(atsc_amd.FIR_FFT)
    enc.encode_batch(ts.data_ptr(), nf, None, iq.data_ptr(), None)
    torch.cuda.synchronize()
    p, n, _ = enc.stage_buffer(atsc_amd.STAGE_TRELLIS_MAP)
    assert p and n == sym_bytes
    assert held < 4 * sym_bytes   # (il, parity, oscillator table ... but not the 67 MB symbol buffer on top of slack)


def test_streaming_call_is_ordered_behind_a_batch_on_another_stream(torch_cuda):
    """atsc_hip_process runs on private streams: a batch queued for the same handle on the caller's own non-blocking
    stream just before must still come first (the interleaver tail, trellis state and oscillator it leaves are the
    streaming call's history)."""
    torch = torch_cuda
    nf = 3
    ts = ts_mt(515, nf)
    want = Oracle().process_fields(ts, nf, want=("iq",))["iq"].reshape(nf, -1)
    enc = atsc_amd.Encoder(channels=1, max_fields=2)
    side = torch.cuda.Stream()
    d_ts = torch.from_numpy(ts[:2 * FIELD_TS_BYTES]).cuda()
    d_iq = torch.empty(2 * FIELD_SYMBOLS, dtype=torch.complex64, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
        for _ in range(8):
            big.normal_()          # keeps the side stream busy: the batch behind it has not started when process() is called
        enc.encode_batch(d_ts.data_ptr(), 2, None, d_iq.data_ptr(), side.cuda_stream)
    got = []

    def fn(ptr, nbytes):
        import ctypes as C
        got.append(np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=np.float32).copy())

    enc.process(ts[2 * FIELD_TS_BYTES:], 312, fn)
    torch.cuda.synchronize()
    assert len(got) == 1
    assert np.abs(got[0] - want[2]).max() <= IQ_TOL
    first = torch.view_as_real(d_iq).cpu().numpy().reshape(2, -1)
    assert np.abs(first - want[:2]).max() <= IQ_TOL


# ----------------------------------------------------------------------------------------------------------------------
# the wide form (fir_w.hip): oscillator folded into the tap operand, 64 channels per wave


@pytest.mark.parametrize("C_,nf", [(64, 3), (128, 1)])
def test_wide_form_symbols_and_iq(torch_cuda, C_, nf):
    """ATSC_HIP_FIR_WIDE regenerates the symbols inside the FIR kernel like FUSED (lane = channel).  Its symbol tap must
    equal k_trellis_map's output bit for bit on every channel, its IQ must agree with the fp32 (FFT) form within a few
    1e-6 on every channel and field, the IQ must not depend on whether the tap is taken, and nothing may be written past
    the buffers; two consecutive batches exercise the carried state.  The first, a middle and the last channel are checked
    against the oracle.  Handles whose channel count is not a multiple of 64 never run it."""
    part = atsc_amd.Encoder(channels=52, max_fields=1)
    assert part.fir_form() == atsc_amd.FIR_FUSED and _code(part.set_fir_form, atsc_amd.FIR_WIDE) == atsc_amd.ERR_ARG
    assert atsc_amd.Encoder(channels=C_, max_fields=1).fir_form() == atsc_amd.FIR_FUSED   # AUTO never picks the wide form
    from oracle_lib import ts_mix
    torch = torch_cuda
    per = [ts_mix(900 + c, 2 * nf) for c in range(C_)]
    st = torch.cuda.current_stream().cuda_stream
    n = C_ * nf * FIELD_SYMBOLS
    outs = {}
    for name, form, tap in (("fft", atsc_amd.FIR_FFT, True), ("wide", atsc_amd.FIR_WIDE, True), ("wide_notap", atsc_amd.FIR_WIDE, False)):
        enc = atsc_amd.Encoder(channels=C_, max_fields=nf)
        enc.set_fir_form(form)
        if name != "fft":
            assert enc.fir_form() == atsc_amd.FIR_WIDE
        res = []
        for b in range(2):
            ts = np.concatenate([p[b * nf * FIELD_TS_BYTES:(b + 1) * nf * FIELD_TS_BYTES] for p in per])
            d_ts = torch.from_numpy(ts).cuda()
            d_sym = torch.full((n + 4096,), 0xEE, dtype=torch.uint8, device="cuda")
            d_iq = torch.full((2 * n + 4096,), float("nan"), dtype=torch.float32, device="cuda")
            enc.encode_batch(d_ts.data_ptr(), nf, d_sym.data_ptr() if tap else None, d_iq.data_ptr(), st)
            torch.cuda.synchronize()
            assert bool((d_sym[n:] == 0xEE).all()) and bool(torch.isnan(d_iq[2 * n:]).all()), "wrote past the buffers"
            res.append((d_sym[:n].view(C_, nf, -1), d_iq[:2 * n].view(C_, nf, -1)))
        outs[name] = res
    worst = 0.0
    for b in range(2):
        s_ref, q_ref = outs["fft"][b]
        s_w, q_w = outs["wide"][b]
        bad = (s_ref != s_w).nonzero()
        assert bad.numel() == 0, f"batch {b}: {bad.shape[0]} symbol mismatches, first (ch, field, pos) {bad[:5].tolist()}"
        assert not bool(torch.isnan(q_w).any()), f"batch {b}: IQ samples never written"
        assert torch.equal(q_w.view(torch.int32), outs["wide_notap"][b][1].view(torch.int32)), "the symbol tap changes the IQ"
        worst = max(worst, float((q_w - q_ref).abs().max()))
    print(f"wide form vs fp32 form, {C_} ch x {nf} fields: max |dIQ| = {worst:.3e}")
    assert worst <= 4e-6
    for c in (0, C_ // 2 + 1, C_ - 1):
        want = Oracle().process_fields(per[c], 2 * nf, want=("sym", "iq"))
        got_sym = np.concatenate([outs["wide"][b][0][c].reshape(-1).cpu().numpy() for b in range(2)])
        got_iq = np.concatenate([outs["wide"][b][1][c].reshape(-1).cpu().numpy() for b in range(2)])
        assert np.array_equal(got_sym, want["sym"])
        assert np.abs(got_iq - want["iq"]).max() <= IQ_TOL


def _model_field(sym_row, t_field, taps):
    """float64: sum_k taps[k] z[i + k], z = fl32(level * T) per component (offset.h:24-31), on the field followed by the copy of
    its first segment (field_sync.h:16)."""
    s = sym_row.astype(np.float64)
    lv = 2.0 * np.concatenate([s, s[:832]]) - 5.75
    n = lv.size
    zr = (lv * t_field[:n, 0].astype(np.float64)).astype(np.float32).astype(np.float64)
    zi = (lv * t_field[:n, 1].astype(np.float64)).astype(np.float32).astype(np.float64)
    t = taps.astype(np.float64)
    return np.stack([np.correlate(zr, t, "valid")[:FIELD_SYMBOLS], np.correlate(zi, t, "valid")[:FIELD_SYMBOLS]], axis=1)


@pytest.mark.parametrize("C_,field,up", [(16, 0, 1), (64, 63, 1), (16, 63, 2)])
def test_matrix_core_fir_against_chosen_symbols_field_and_phase(torch_cuda, C_, field, up):
    """VERDICT r2 item 5.  k_mix_fir_mm leaves low-order fp16 products out of its outer K-steps; scripts/fir_error_budget.py
    bounds every term of what that and the fp16 pieces can cost.  Here (a) the budget is recomputed from the operand READ BACK
    from the device (atsc_hip_get_table id 8 with a handle), and (b) the symbols are chosen against the kernel -- a float64
    restatement of its omitted terms (tests/fir_adversary.py) picks, for a target every 161 outputs, the level of every input
    of the outer K-steps that pushes the error one way -- at field 63 (T_63, not T_0), on a 64-channel handle (four wave
    groups), and for the half-symbol operand of the 2x form (odd samples).  The result stays inside 1e-5 of the float64
    model everywhere, inside 1e-5 of the oracle on the channels compared, and the error AT the targets is the one aimed for
    (the restatement describes the kernel)."""
    import fir_adversary as adv
    torch = torch_cuda
    enc = atsc_amd.Encoder(channels=C_, max_fields=1)
    enc.set_fir_form(atsc_amd.FIR_MFMA)
    enc.set_upsample(up)
    # (a) the operand the kernels of this device multiply by = the host-built one, bit for bit; the budget from those bits
    dev_bits = atsc_amd.get_table(8, enc.h)
    assert np.array_equal(dev_bits, atsc_amd.get_table(8)), "device copy of the tap operand differs from the host build"
    assert np.array_equal(atsc_amd.get_table(9, enc.h), atsc_amd.get_table(9))
    operand = adv.operand(enc.h)
    for ph in range(up):
        total, terms = adv.feb.fused_budget(ph, handle=enc.h, fields=max(field + 1, 2), c=1.0)
        print(f"phase {ph}: budget from the device operand {total:.3e}", {k[:1]: f"{v:.2e}" for k, v in terms.items()})
        assert total < IQ_TOL
    # the oscillator of this field from the oracle's own recursion (mix_fir on dummy symbols moves it on)
    o = Oracle()
    if field:
        o.mix_fir(np.zeros(field * FIELD_SYMBOLS, np.uint8), field)
    t_field = o.osc_table().reshape(-1, 2).astype(np.float32)
    rng = np.random.default_rng(5 + field + up)
    sym = rng.integers(0, 8, (C_, FIELD_SYMBOLS), dtype=np.uint8)
    targets = []
    for c in range(C_):
        for n_t, i in enumerate(range(1000 + 7 * c, FIELD_SYMBOLS - 400, 161 * max(1, C_ // 16))):
            comp, ph = (c + n_t) & 1, (n_t >> 1) % up
            sign = 1.0 if (n_t >> 2) & 1 else -1.0
            aimed = adv.choose(sym, c, i // 32, i % 32, comp, t_field, ph, operand, sign)
            targets.append((c, i, comp, ph, aimed))
    if field:
        enc.seek(field)
    d_sym = torch.from_numpy(sym).cuda()
    d_iq = torch.empty((C_, up * FIELD_SYMBOLS), dtype=torch.complex64, device="cuda")
    enc.mix_fir(d_sym.data_ptr(), 1, d_iq.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = d_iq.cpu().numpy().view(np.float32).reshape(C_, FIELD_SYMBOLS, up, 2).astype(np.float64)
    taps = [atsc_amd.get_table(0)[:111], atsc_amd.get_table(7)[:112]]
    t = np.array([(c, i, comp, ph) for c, i, comp, ph, _ in targets])
    aimed = np.array([a for *_, a in targets])
    worst = 0.0
    err_at = np.zeros(len(targets))
    for c in range(C_):
        for ph in range(up):
            diff = got[c, :, ph, :] - _model_field(sym[c], t_field, taps[ph])
            worst = max(worst, float(np.abs(diff).max()))
            sel = np.nonzero((t[:, 0] == c) & (t[:, 3] == ph))[0]
            err_at[sel] = -diff[t[sel, 1], t[sel, 2]]     # `aimed` is exact minus kernel
    print(f"C={C_} field={field} up={up}: max |error| vs the float64 model {worst:.3e}; at the {len(targets)} targets: aimed "
          f"{np.abs(aimed).max():.3e} max / {np.abs(aimed).mean():.3e} mean, seen {np.abs(err_at).max():.3e} max, "
          f"|seen - aimed| {np.abs(err_at - aimed).max():.3e} max")
    assert worst <= IQ_TOL
    assert np.abs(aimed).max() > 1.5e-6                     # the vector does what it is built for (random data: 1.5e-6 over 1e9 samples)
    assert np.abs(err_at - aimed).max() < 2.0e-6            # and the restatement is the kernel: the rest is the inner K-steps' residue
    assert np.corrcoef(err_at, aimed)[0, 1] > 0.8
    # against the oracle's fp32 FIR (the pinned half: phase 0) on two channels
    for c in sorted({0, C_ - 1}):
        oc = Oracle()
        if field:
            oc.mix_fir(np.zeros(field * FIELD_SYMBOLS, np.uint8), field)
        want = oc.mix_fir(sym[c], 1).reshape(FIELD_SYMBOLS, 2)
        assert np.abs(got[c, :, 0, :] - want).max() <= IQ_TOL


def test_iq_loopback_full_size_symbol_rate(torch_cuda):
    """the loop-back closed at the IQ, through the mixer and the FIR.  BASELINE config 4's per-GPU shard, 64
    DISTINCT random streams x 64 fields through the default (fused matrix-core) path; every channel's complex64 IQ goes
    through tests/loopback.py's receiver -- its own 255-tap matched filter, an ideal double-precision carrier, per-field
    gain from the pilot, slicer -- and the existing de-interleave / RS / de-randomize chain, and gives back the TS that went in
    with every RS codeword valid; the soft levels sit within 0.1 of a level step of the 8 levels.  On four channels the
    Viterbi decoder then takes the same IQ with noise added to it.  No oracle, no product table."""
    from loopback import decode_symbols, receive_iq, viterbi_symbols
    torch = torch_cuda
    C_, nf = 64, 64
    g = torch.Generator(device="cuda")
    g.manual_seed(4242)
    ts = torch.randint(0, 256, (C_, nf * 312, 188), dtype=torch.uint8, device="cuda", generator=g)
    ts[:, :, 0] = 0x47
    enc = atsc_amd.Encoder(channels=C_, max_fields=nf)
    n = nf * FIELD_SYMBOLS
    d_iq = torch.empty((C_, n), dtype=torch.complex64, device="cuda")
    enc.encode_batch(ts.data_ptr(), nf, None, d_iq.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert enc.fir_form() == atsc_amd.FIR_FUSED
    worst_soft = 0.0
    for c in range(C_):
        lv = receive_iq(d_iq[c], nf)
        hard = lv.round().clamp(0, 7)
        worst_soft = max(worst_soft, float((lv - hard)[64:n - 400].abs().max()))
        payload, worst = decode_symbols(hard.to(torch.uint8), nf)
        assert worst == 0, f"channel {c}: RS syndromes not zero"
        assert bool((payload == ts[c, :(nf - 1) * 312, 1:]).all()), f"channel {c}: TS recovered from the IQ differs"
    print(f"soft levels: at most {worst_soft:.3f} of a level step from a level")
    assert worst_soft < 0.15
    k = 4
    for c in (0, 21, 42, 63):
        rx = d_iq[c, :k * FIELD_SYMBOLS]
        g.manual_seed(100 + c)
        noise = torch.randn((2, rx.numel()), generator=g, dtype=torch.float32, device="cuda")
        rx = rx + 0.03 * torch.complex(noise[0], noise[1])           # 0.24 of a level step rms at the slicer: it errs on 3 % of the symbols
        lv = receive_iq(rx, k)
        payload, worst = decode_symbols(viterbi_symbols(lv, k), k)
        assert worst == 0 and bool((payload == ts[c, :(k - 1) * 312, 1:]).all()), f"channel {c}: noisy IQ not recovered"


def test_iq_loopback_odd_samples_of_the_2x_form(torch_cuda):
    """The same from the interpolating form, using ONLY the odd samples y2[2 i + 1] (the half-symbol operand: the samples the
    reference has no counterpart for, include/atsc_hip.h atsc_hip_set_upsample): 16 streams x 4 fields; the receiver's pulse is
    sampled half a sample off, which puts its output back on the symbol instants; Viterbi, de-interleave, RS -> the TS that
    went in.  A wrong half-symbol operand (pulse, scale, alignment) cannot pass: the symbols would not be there."""
    from loopback import decode_symbols, receive_iq, viterbi_symbols
    torch = torch_cuda
    C_, nf = 16, 4
    g = torch.Generator(device="cuda")
    g.manual_seed(777)
    ts = torch.randint(0, 256, (C_, nf * 312, 188), dtype=torch.uint8, device="cuda", generator=g)
    ts[:, :, 0] = 0x47
    enc = atsc_amd.Encoder(channels=C_, max_fields=nf)
    enc.set_upsample(2)
    n = nf * FIELD_SYMBOLS
    d_iq = torch.empty((C_, 2 * n), dtype=torch.complex64, device="cuda")
    enc.encode_batch(ts.data_ptr(), nf, None, d_iq.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    worst_soft = 0.0
    for c in range(C_):
        lv = receive_iq(d_iq[c, 1::2].contiguous(), nf, half_symbol_late=True)
        worst_soft = max(worst_soft, float((lv - lv.round().clamp(0, 7))[64:n - 400].abs().max()))
        payload, worst = decode_symbols(viterbi_symbols(lv, nf), nf)
        assert worst == 0, f"channel {c}: RS syndromes not zero"
        assert bool((payload == ts[c, :(nf - 1) * 312, 1:]).all()), f"channel {c}: TS recovered from the odd samples differs"
        if c == 0:   # and the even samples are the symbol-rate signal
            p0, w0 = decode_symbols(receive_iq(d_iq[c, 0::2].contiguous(), nf).round().clamp(0, 7).to(torch.uint8), nf)
            assert w0 == 0 and bool((p0 == ts[c, :(nf - 1) * 312, 1:]).all())
    print(f"odd samples: soft levels at most {worst_soft:.3f} of a level step from a level")
    assert worst_soft < 0.15


def test_multi_stream_front_32_ragged_streams_against_the_oracle(torch_cuda):
    """atsc_hip_process_multi -- 32 DISTINCT streams on one 32-channel handle, each fed through the
    reference's process() contract in ragged chunk sizes and in a shuffled order; every stream's callbacks arrive in field
    order with that stream's IQ (three streams against the oracle's fp32 IQ, every stream against a batch call on the same
    TS, bit for bit), the handle runs the fused matrix-core form, nothing is left in flight, partial fields stay buffered,
    a stream that runs too far ahead is refused without being copied."""
    torch = torch_cuda
    K, nf, F = 32, 5, 2
    per = [ts_mt(3000 + c, nf) for c in range(K)]
    enc = atsc_amd.Encoder(channels=K, max_fields=F)
    got = [[] for _ in range(K)]
    order = []

    def sink(c):
        def fn(ptr, nbytes):
            assert nbytes == FIELD_SYMBOLS * 8
            got[c].append(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(FIELD_SYMBOLS * 2,)).copy())
            order.append(c)
        return fn

    rng = np.random.default_rng(11)
    sent = [0] * K
    total = nf * 312
    fields_seen = 0
    while any(s < total for s in sent):
        c = int(rng.choice([i for i in range(K) if sent[i] < total]))
        cnt = int(min(total - sent[c], rng.choice([1, 17, 100, 311, 312, 313, 700])))
        # never further ahead of the slowest stream than the front allows
        if (sent[c] + cnt) // 312 - min(sent) // 312 > atsc_amd.MULTI_MAX_AHEAD * F:
            continue
        before = min(sent) // 312
        enc.process_multi(c, per[c][sent[c] * 188:(sent[c] + cnt) * 188], cnt, sink(c))
        sent[c] += cnt
        after = min(sent) // 312
        # every field all streams have completed has been delivered to every stream when the call returns
        assert all(len(g_) == after for g_ in got), (before, after, [len(g_) for g_ in got][:4])
        fields_seen = after
        assert enc.multi_pending(c) == sent[c] - after * 312
    assert fields_seen == nf and enc.fir_form() == atsc_amd.FIR_FUSED
    # field-major delivery: within a batch the streams are served 0..K-1 per field
    assert order[:K] == list(range(K))
    # bit-identical to one batch call over the same TS on a fresh handle (same form, same grouping)
    ref = atsc_amd.Encoder(channels=K, max_fields=nf)
    d_ts = torch.from_numpy(np.concatenate(per)).cuda()
    d_iq = torch.empty((K, nf * FIELD_SYMBOLS), dtype=torch.complex64, device="cuda")
    ref.encode_batch(d_ts.data_ptr(), nf, None, d_iq.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want_all = d_iq.cpu().numpy().view(np.float32).reshape(K, nf, FIELD_SYMBOLS * 2)
    for c in range(K):
        assert np.array_equal(np.stack(got[c]).view(np.uint32), want_all[c].view(np.uint32)), f"stream {c} differs from the batch call"
    for c in (0, 13, K - 1):
        want = Oracle().process_fields(per[c], nf, want=("iq",))["iq"].reshape(nf, -1)
        assert np.abs(np.stack(got[c]) - want).max() <= IQ_TOL
    # too far ahead: refused, nothing copied
    big = ts_mt(1, atsc_amd.MULTI_MAX_AHEAD * F + 2)
    with pytest.raises(atsc_amd.AtscHipError) as ei:
        enc.process_multi(3, big, big.size // 188, sink(3))
    assert ei.value.code == atsc_amd.ERR_TOO_LARGE and enc.multi_pending(3) == 0
    # argument checks; a 1-channel handle may use either front, but not both with bytes buffered
    with pytest.raises(atsc_amd.AtscHipError):
        enc.process_multi(K, per[0], 1, sink(0))
    one = atsc_amd.Encoder(channels=1, max_fields=1)
    one.process_multi(0, per[0][:188 * 10], 10, sink(0))
    with pytest.raises(atsc_amd.AtscHipError) as ei:
        one.process(per[0][:188], 1, lambda *a: None)
    assert ei.value.code == atsc_amd.ERR_STATE


def test_null_driver_streams_option(torch_cuda, tmp_path):
    """atsc_encoder_null --streams 16: the reference-shaped C++ group adapter (atsc::atsc_encoder_group, one 16-channel handle)
    end to end -- 16 x the fields, the fused form, and the CS16 byte count."""
    import json
    import os
    import subprocess
    exe = os.path.join(atsc_amd.ROOT, "atsc_amd", "bin", "atsc_encoder_null")
    assert os.path.exists(exe), "driver not built (atsc_amd/csrc/Makefile)"
    path = tmp_path / "in.ts"
    path.write_bytes(ts_mt(8, 9).tobytes())
    out = subprocess.run([exe, str(path), "2", "0", "cs16", "--streams", "16"], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout)
    assert r["streams"] == 16 and r["fir_form"] == atsc_amd.FIR_FUSED
    assert r["fields"] == 16 * 9 and r["iq_bytes"] == 16 * 9 * FIELD_SYMBOLS * 4


def test_timing_select_times_the_chosen_stage_only(torch_cuda):
    """atsc_hip_timing_select: with the mixer + FIR stage alone selected a batch records that kernel's two events (on the fused
    path they ride on the kernel's own dispatch) and every other stage reads 0; with everything selected the front stages have
    times again; a handle without the timing flag refuses the call."""
    torch = torch_cuda
    C_, nf = 16, 2
    ts = torch.from_numpy(np.concatenate([ts_mt(40 + c, nf) for c in range(C_)])).cuda()
    iq = torch.empty((C_, nf * FIELD_SYMBOLS), dtype=torch.complex64, device="cuda")
    enc = atsc_amd.Encoder(channels=C_, max_fields=nf, timing=True)
    stream = torch.cuda.current_stream().cuda_stream
    enc.timing_reserve(4)
    enc.timing_select(("mix_fir",))
    enc.encode_batch(ts.data_ptr(), nf, None, iq.data_ptr(), stream)
    enc.timing_select(None)
    enc.encode_batch(ts.data_ptr(), nf, None, iq.data_ptr(), stream)
    torch.cuda.synchronize()
    only, every = enc.batch_timings(0), enc.batch_timings(1)
    assert enc.fir_form() == atsc_amd.FIR_FUSED
    assert only["mix_fir"] > 0 and all(v == 0 for k, v in only.items() if k != "mix_fir"), only
    assert every["mix_fir"] > 0 and every["rand_rs"] > 0 and every["interleave"] > 0 and every["osc_table"] > 0, every
    assert abs(only["mix_fir"] - every["mix_fir"]) < 0.5 * every["mix_fir"]
    plain = atsc_amd.Encoder(channels=1, max_fields=1)
    with pytest.raises(atsc_amd.AtscHipError) as ei:
        plain.timing_select(("mix_fir",))
    assert ei.value.code == atsc_amd.ERR_STATE
