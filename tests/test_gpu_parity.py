"""GPU parity tests (run on the MI355X box with -m gpu).  Everything goes through the C-ABI
(atsc_amd/lib/libatsc_hip.so via ctypes); the oracle (oracle/atsc_oracle.c) is only the checker.

Bars (north_star): coded bytes, interleaved bytes and 8-level symbols bit-exact; IQ max abs error
<= 1e-5 of full scale (full scale = 1.0).
"""
import ctypes as C
import hashlib

import numpy as np
import pytest

import atsc_amd
from atsc_amd import FIELD_CODED_BYTES, FIELD_SYMBOLS, FIELD_TS_BYTES, INTERLEAVER_TAIL
from oracle_lib import Oracle, Ref, ref_available, ts_mix, ts_mt, ts_null

pytestmark = pytest.mark.gpu
IQ_TOL = 1e-5  # of full scale 1.0


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def d2h(ptr, nbytes):
    hip = C.CDLL("libamdhip64.so")
    out = np.empty(nbytes, np.uint8)
    rc = hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(nbytes), 2)
    assert rc == 0
    return out


def run_device(torch, enc, ts, channels, fields, want_iq=True):
    """Device-resident batch through atsc_hip_encode_batch; returns (sym, iq) numpy."""
    d_ts = torch.from_numpy(np.ascontiguousarray(ts)).cuda()
    d_sym = torch.empty(channels * fields * FIELD_SYMBOLS, dtype=torch.uint8, device="cuda")
    d_iq = torch.empty(channels * fields * FIELD_SYMBOLS, dtype=torch.complex64, device="cuda") if want_iq else None
    enc.encode_batch(d_ts.data_ptr(), fields, d_sym.data_ptr(), d_iq.data_ptr() if want_iq else None,
                     torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_sym.cpu().numpy(), (d_iq.cpu().numpy() if want_iq else None)


INPUTS = {"null": lambda nf: ts_null(nf), "mt1234": lambda nf: ts_mt(1234, nf), "mix": lambda nf: ts_mix(5, nf)}


@pytest.mark.parametrize("name", list(INPUTS))
def test_all_stages_4_fields(torch_cuda, name):
    """Every stage tap against the oracle on 4 fields (first field has zero history, odd/even syncs)."""
    nf = 4
    ts = INPUTS[name](nf)
    want = Oracle().process_fields(ts, nf)
    enc = atsc_amd.Encoder(channels=1, max_fields=nf, taps=True)
    sym, iq = run_device(torch_cuda, enc, ts, 1, nf)
    p, _, _ = enc.stage_buffer(atsc_amd.STAGE_RAND_RS)
    coded = d2h(p, INTERLEAVER_TAIL + nf * FIELD_CODED_BYTES)
    assert np.array_equal(coded[INTERLEAVER_TAIL:], want["enc"]), "RS-coded bytes differ"
    # the saved stream state carries the tail of the last field
    st = enc.export_state(0)
    assert np.array_equal(np.frombuffer(bytes(st.coded_tail), np.uint8), want["enc"][-INTERLEAVER_TAIL:])
    p, _, _ = enc.stage_buffer(atsc_amd.STAGE_INTERLEAVE)
    il = d2h(p, nf * FIELD_CODED_BYTES)
    assert np.array_equal(il, want["il"]), "interleaved bytes differ"
    bad = np.flatnonzero(sym != want["sym"])
    assert bad.size == 0, f"{bad.size} symbol mismatches, first at {bad[:8]}"
    err = np.abs(iq.view(np.float32) - want["iq"]).max()
    print(f"{name}: IQ max abs err {err:.3e}")
    assert err <= IQ_TOL


def test_golden_hashes_on_gpu(torch_cuda, golden):
    """Symbols of the GPU path hash to the values recorded from the compiled reference."""
    g, w = golden
    for name, ts in (("null4", ts_null(4)), ("mt1234_4", ts_mt(1234, 4))):
        enc = atsc_amd.Encoder(channels=1, max_fields=4)
        sym, iq = run_device(torch_cuda, enc, ts, 1, 4)
        assert sha(sym) == g["cases"][name]["sha256"]["sym"]
        err = np.abs(iq[w[name + "_iq_idx"]] - w[name + "_iq"]).max()
        assert err <= IQ_TOL


@pytest.mark.parametrize("name", list(INPUTS))
def test_config2_64_field_symbols(torch_cuda, name):
    """BASELINE config 2: 1 channel x 64 fields, 0 symbol mismatches over 64*260416 positions."""
    nf = 64
    ts = INPUTS[name](nf)
    want = Oracle().symbols(ts, nf)
    enc = atsc_amd.Encoder(channels=1, max_fields=nf)
    sym, _ = run_device(torch_cuda, enc, ts, 1, nf, want_iq=False)
    assert int((sym != want).sum()) == 0


def test_config3_64_field_iq(torch_cuda, golden):
    """BASELINE config 3: full chain IQ on 64 fields within 1e-5 of full scale (oscillator recursion tracked)."""
    nf = 64
    ts = ts_mt(1234, nf)
    # the compiled reference itself where oracle/_ref travelled with the tree, else the restatement (which the golden
    # vectors pin to the reference): either way the 64-field public-API hash of golden.json is checked first, so the
    # "expected" side is the reference's output, and the test says which checker produced it
    if ref_available():
        which, want = "compiled reference (oracle/_ref)", Ref().process_fields(ts, nf, want=("iq",))["iq"]
    else:
        which, want = "C restatement (oracle/atsc_oracle.c)", Oracle().process_fields(ts, nf, want=("iq",))["iq"]
    enc = atsc_amd.Encoder(channels=1, max_fields=nf)
    _, iq = run_device(torch_cuda, enc, ts, 1, nf)
    err = np.abs(iq.view(np.float32) - want)
    print(f"64-field IQ against the {which}: max abs err {err.max():.3e}, last field {err[-2 * FIELD_SYMBOLS:].max():.3e}")
    assert err.max() <= IQ_TOL, which


def test_multichannel_and_batch_split(torch_cuda):
    """3 channels x 5 fields in one batch == each channel alone == the same stream fed as 2+3 fields."""
    C_, nf = 3, 5
    per = [ts_mt(1234 + c, nf) for c in range(C_)]
    ts = np.concatenate(per)
    enc = atsc_amd.Encoder(channels=C_, max_fields=nf)
    sym, iq = run_device(torch_cuda, enc, ts, C_, nf)
    sym = sym.reshape(C_, -1)
    iq = iq.reshape(C_, -1)
    for c in range(C_):
        want = Oracle().process_fields(per[c], nf, want=("sym", "iq"))
        assert np.array_equal(sym[c], want["sym"])
        assert np.abs(iq[c].view(np.float32) - want["iq"]).max() <= IQ_TOL
    # split batches: state (interleaver tail, carries, reserved symbols, field parity, oscillator) carried
    enc2 = atsc_amd.Encoder(channels=C_, max_fields=3)
    a = np.concatenate([p[:2 * FIELD_TS_BYTES] for p in per])
    b = np.concatenate([p[2 * FIELD_TS_BYTES:] for p in per])
    s1, q1 = run_device(torch_cuda, enc2, a, C_, 2)
    s2, q2 = run_device(torch_cuda, enc2, b, C_, 3)
    s = np.concatenate([s1.reshape(C_, -1), s2.reshape(C_, -1)], axis=1)
    q = np.concatenate([q1.reshape(C_, -1), q2.reshape(C_, -1)], axis=1)
    assert np.array_equal(s, sym)
    assert np.array_equal(q.view(np.uint32), iq.view(np.uint32)), "batch split changed IQ bits"


def test_channel_groups_11_channels(torch_cuda):
    """11 channels x 2 fields: the FIR walks channels in tasks of 8 and in pairs, so 11 = 8 + 3 exercises a full task,
    a short task and an odd pair (whose second stream is a dummy); every channel against the oracle."""
    C_, nf = 11, 2
    per = [ts_mix(77 + c, nf) for c in range(C_)]
    enc = atsc_amd.Encoder(channels=C_, max_fields=nf)
    sym, iq = run_device(torch_cuda, enc, np.concatenate(per), C_, nf)
    sym = sym.reshape(C_, -1)
    iq = iq.reshape(C_, -1)
    for c in range(C_):
        want = Oracle().process_fields(per[c], nf, want=("sym", "iq"))
        assert np.array_equal(sym[c], want["sym"]), f"channel {c}"
        assert np.abs(iq[c].view(np.float32) - want["iq"]).max() <= IQ_TOL, f"channel {c}"


def test_full_size_batch_channel_independence(torch_cuda):
    """BASELINE config 4 per-GPU shard (64 channels x 64 fields) through a size-independent property: channels are
    independent and deterministic, so 64 copies of one stream give 64 bit-identical outputs; their symbols equal that
    stream encoded alone, whose IQ agrees
    within the tolerance; and channel 0 is checked against the oracle on its first and last fields."""
    torch = torch_cuda
    C_, nf = 64, 64
    one = ts_mt(4321, nf)
    d_one = torch.from_numpy(one).cuda()
    d_ts = d_one.repeat(C_).contiguous()
    enc = atsc_amd.Encoder(channels=C_, max_fields=nf)
    n = nf * FIELD_SYMBOLS
    d_sym = torch.empty(C_ * n, dtype=torch.uint8, device="cuda")
    d_iq = torch.empty(C_ * n, dtype=torch.complex64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    enc.encode_batch(d_ts.data_ptr(), nf, d_sym.data_ptr(), d_iq.data_ptr(), st)
    torch.cuda.synchronize()
    solo = atsc_amd.Encoder(channels=1, max_fields=nf)
    s_sym = torch.empty(n, dtype=torch.uint8, device="cuda")
    s_iq = torch.empty(n, dtype=torch.complex64, device="cuda")
    solo.encode_batch(d_one.data_ptr(), nf, s_sym.data_ptr(), s_iq.data_ptr(), st)
    torch.cuda.synchronize()
    sym2 = d_sym.view(C_, n)
    iqf = torch.view_as_real(d_iq).view(C_, 2 * n)
    assert bool((sym2 == s_sym.unsqueeze(0)).all()), "channels differ in symbols"
    assert bool((iqf.view(torch.int32) == iqf[0:1].view(torch.int32)).all()), "channels differ in IQ bits"
    assert float((iqf[0] - torch.view_as_real(s_iq).view(2 * n)).abs().max()) <= IQ_TOL, "batch and solo forms disagree"
    want = Oracle().process_fields(one, nf, want=("sym", "iq"))
    assert np.array_equal(s_sym.cpu().numpy(), want["sym"])
    got = iqf[0].cpu().numpy()
    for f in (0, nf - 1):
        sl = slice(2 * f * FIELD_SYMBOLS, 2 * (f + 1) * FIELD_SYMBOLS)
        assert np.abs(got[sl] - want["iq"][sl]).max() <= IQ_TOL


def test_full_size_loopback_roundtrip(torch_cuda):
    """BASELINE config 4 per-GPU shard with 64 DISTINCT random streams x 64 fields: every channel's symbols are decoded
    back on the GPU by tests/loopback.py (independent demap / inverse precoder / de-interleave / RS syndromes /
    de-randomize) and must give the TS that went in, with every RS codeword valid; every data symbol satisfies the
    trellis code's parity check; and four fields of one channel, received with Gaussian noise that makes the plain slicer
    err on about 4 % of the symbols (0.25 level steps rms: the code's error events, at distance 3, are then
    a 6-sigma matter), come back intact through the verifier's Viterbi decoder.  No oracle involved."""
    from loopback import decode_symbols, trellis_violations, viterbi_symbols
    torch = torch_cuda
    C_, nf = 64, 64
    g = torch.Generator(device="cuda")
    g.manual_seed(2025)
    ts = torch.randint(0, 256, (C_, nf * 312, 188), dtype=torch.uint8, device="cuda", generator=g)
    ts[:, :, 0] = 0x47
    enc = atsc_amd.Encoder(channels=C_, max_fields=nf)
    n = nf * FIELD_SYMBOLS
    d_sym = torch.empty(C_ * n, dtype=torch.uint8, device="cuda")
    enc.encode_batch(ts.data_ptr(), nf, d_sym.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for c in range(C_):
        payload, worst = decode_symbols(d_sym[c * n:(c + 1) * n], nf)
        assert worst == 0, f"channel {c}: RS syndromes not zero"
        assert bool((payload == ts[c, :(nf - 1) * 312, 1:]).all()), f"channel {c}: decoded TS differs"
        assert trellis_violations(d_sym[c * n:(c + 1) * n], nf) == 0, f"channel {c}: not a trellis code sequence"
    c, k = 17, 4
    clean = d_sym[c * n:c * n + k * FIELD_SYMBOLS]
    g.manual_seed(9)
    rx = clean.double() + 0.25 * torch.randn(clean.shape, generator=g, dtype=torch.float64, device="cuda")
    assert int((rx.round().clamp(0, 7).to(torch.uint8) != clean).sum()) > 0.03 * clean.numel()
    payload, worst = decode_symbols(viterbi_symbols(rx, k), k)
    assert worst == 0
    assert bool((payload == ts[c, :(k - 1) * 312, 1:]).all())


def test_streaming_process_api(torch_cuda):
    """atsc_hip_process: arbitrary packet counts, one synchronous in-order callback per field."""
    nf = 6
    ts = ts_mt(77, nf)
    want = Oracle().process_fields(ts, nf, want=("iq",))["iq"].reshape(nf, -1)
    enc = atsc_amd.Encoder(channels=1, max_fields=2)
    got = []

    def on_field(ptr, nbytes):
        assert nbytes == FIELD_SYMBOLS * 8
        got.append(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(FIELD_SYMBOLS * 2,)).copy())

    counts = []
    npk = nf * 312
    off = 0
    for cnt in (100, 212, 1000, 1, 311, npk - 1624):
        enc.process(ts[off * 188:(off + cnt) * 188], cnt, on_field)
        off += cnt
        counts.append(len(got))
    assert off == npk
    assert counts == [0, 1, 4, 4, 5, 6]
    for f in range(nf):
        assert np.abs(got[f] - want[f]).max() <= IQ_TOL


def test_state_export_import_roundtrip(torch_cuda):
    """A second encoder that imports the state exported after field 3 continues bit-identically."""
    nf = 7
    ts = ts_mix(21, nf)
    full = atsc_amd.Encoder(channels=1, max_fields=nf)
    sym_full, iq_full = run_device(torch_cuda, full, ts, 1, nf)
    a = atsc_amd.Encoder(channels=1, max_fields=4)
    run_device(torch_cuda, a, ts[:3 * FIELD_TS_BYTES], 1, 3)
    st = a.export_state(0)
    assert st.field_index == 3
    # state agrees with the oracle's delay elements
    o = Oracle()
    o.symbols(ts[:3 * FIELD_TS_BYTES], 3)
    os_ = o.state()
    assert list(st.z2) == os_["z2"].tolist() and list(st.z0_next) == os_["d1"].tolist() and list(st.z0_prev) == os_["d2"].tolist()
    b = atsc_amd.Encoder(channels=1, max_fields=4)
    b.import_state(st, 0)
    sym_b, iq_b = run_device(torch_cuda, b, ts[3 * FIELD_TS_BYTES:], 1, 4)
    assert np.array_equal(sym_b, sym_full[3 * FIELD_SYMBOLS:])
    assert np.array_equal(iq_b.view(np.uint32), iq_full[3 * FIELD_SYMBOLS:].view(np.uint32))


def test_span_parity_linearity(torch_cuda):
    """state_out = state_in XOR span_parity, span parity computed from zero trellis state."""
    ts = ts_mt(5, 4)
    cont = atsc_amd.Encoder(channels=1, max_fields=2)
    run_device(torch_cuda, cont, ts[:2 * FIELD_TS_BYTES], 1, 2, want_iq=False)
    st_in = cont.export_state(0)
    run_device(torch_cuda, cont, ts[2 * FIELD_TS_BYTES:], 1, 2, want_iq=False)
    st_out = cont.export_state(0)
    # cold encoder with the right interleaver history but zero trellis state
    cold = atsc_amd.Encoder(channels=1, max_fields=2)
    st0 = atsc_amd.StreamState()
    C.memmove(C.byref(st0), C.byref(st_in), C.sizeof(st0))
    for i in range(12):
        st0.z2[i] = st0.z0_next[i] = st0.z0_prev[i] = 0
    cold.import_state(st0, 0)
    run_device(torch_cuda, cold, ts[2 * FIELD_TS_BYTES:], 1, 2, want_iq=False)
    par = cold.span_parity(0)
    assert [(st_in.z2[i] ^ par[0][i]) for i in range(12)] == list(st_out.z2)
    assert [(st_in.z0_next[i] ^ par[1][i]) for i in range(12)] == list(st_out.z0_next)
    assert [(st_in.z0_prev[i] ^ par[2][i]) for i in range(12)] == list(st_out.z0_prev)


def test_mix_fir_alone(torch_cuda):
    """atsc_hip_mix_fir on oracle symbols == oracle mixer+FIR (isolates the float stage)."""
    nf = 2
    ts = ts_mt(3, nf)
    o = Oracle()
    d = o.process_fields(ts, nf, want=("sym", "iq"))
    enc = atsc_amd.Encoder(channels=1, max_fields=nf)
    d_sym = torch_cuda.from_numpy(d["sym"]).cuda()
    d_iq = torch_cuda.empty(nf * FIELD_SYMBOLS, dtype=torch_cuda.complex64, device="cuda")
    enc.mix_fir(d_sym.data_ptr(), nf, d_iq.data_ptr(), torch_cuda.cuda.current_stream().cuda_stream)
    torch_cuda.cuda.synchronize()
    err = np.abs(d_iq.cpu().numpy().view(np.float32) - d["iq"]).max()
    assert err <= IQ_TOL
    # oscillator state after 2 fields follows the reference recursion bit-for-bit
    p, n, _ = enc.stage_buffer(atsc_amd.STAGE_MIX_FIR)
    osc = d2h(p, n).view(np.float32)
    assert np.array_equal(osc.view(np.uint32), o.osc_table().view(np.uint32))


def test_errors(torch_cuda):
    enc = atsc_amd.Encoder(channels=1, max_fields=2)
    d_ts = torch_cuda.zeros(3 * FIELD_TS_BYTES, dtype=torch_cuda.uint8, device="cuda")
    with pytest.raises(atsc_amd.AtscHipError):
        enc.encode_batch(d_ts.data_ptr(), 3)       # more fields than the handle was created for
    with pytest.raises(atsc_amd.AtscHipError):
        enc.encode_batch(None, 1)
    with pytest.raises(atsc_amd.AtscHipError):
        atsc_amd.Encoder(channels=0)
    enc2 = atsc_amd.Encoder(channels=2, max_fields=1)
    with pytest.raises(atsc_amd.AtscHipError):
        enc2.process(np.zeros(188, np.uint8), 1, lambda *a: None)  # streaming path is single-channel
    # the symbol stage tap is only there when the last batch wrote the internal symbol buffer
    torch = torch_cuda
    st = torch.cuda.current_stream().cuda_stream
    d_iq = torch.empty(2 * FIELD_SYMBOLS, dtype=torch.complex64, device="cuda")
    enc.encode_batch(d_ts.data_ptr(), 2, None, d_iq.data_ptr(), st)
    p, nbytes, _ = enc.stage_buffer(atsc_amd.STAGE_TRELLIS_MAP)
    assert p and nbytes == 2 * FIELD_SYMBOLS
    d_sym = torch.empty(2 * FIELD_SYMBOLS, dtype=torch.uint8, device="cuda")
    enc.encode_batch(d_ts.data_ptr(), 2, d_sym.data_ptr(), d_iq.data_ptr(), st)
    with pytest.raises(atsc_amd.AtscHipError):
        enc.stage_buffer(atsc_amd.STAGE_TRELLIS_MAP)          # symbols went to the caller's buffer
    enc16 = atsc_amd.Encoder(channels=32, max_fields=1)
    d_ts16 = torch.zeros(32 * FIELD_TS_BYTES, dtype=torch.uint8, device="cuda")
    d_iq16 = torch.empty(32 * FIELD_SYMBOLS, dtype=torch.complex64, device="cuda")
    enc16.encode_batch(d_ts16.data_ptr(), 1, None, d_iq16.data_ptr(), st)
    with pytest.raises(atsc_amd.AtscHipError):
        enc16.stage_buffer(atsc_amd.STAGE_TRELLIS_MAP)        # fused form: no symbols anywhere
    torch.cuda.synchronize()


def test_config5_time_split_equals_continuous(torch_cuda):
    """BASELINE config 5 on one GPU: 4 'ranks' x 3 fields, each a cold handle primed from a one-field TS halo,
    trellis state handed over as the XOR of earlier span parities -> bit-identical symbols AND IQ."""
    from atsc_amd.timesplit import TimeSplitEncoder
    torch = torch_cuda
    R, nf = 4, 3
    ts = ts_mix(99, R * nf)
    cont = atsc_amd.Encoder(channels=1, max_fields=R * nf)
    sym_c, iq_c = run_device(torch, cont, ts, 1, R * nf)
    d_ts = torch.from_numpy(ts).cuda()
    ranks, pars = [], []
    stream = torch.cuda.current_stream().cuda_stream
    for r in range(R):
        t = TimeSplitEncoder(nf)
        halo = d_ts[(r * nf - 1) * FIELD_TS_BYTES:r * nf * FIELD_TS_BYTES] if r else None
        t.prepare(r * nf, halo)
        par = torch.zeros(4, dtype=torch.int32, device="cuda")
        t.enc.encode_begin(d_ts[r * nf * FIELD_TS_BYTES:].data_ptr(), nf, par.data_ptr(), stream)
        ranks.append(t)
        pars.append(par)
    syms, iqs = [], []
    carry = torch.zeros(4, dtype=torch.int32, device="cuda")
    for r in range(R):
        d_sym = torch.empty(nf * FIELD_SYMBOLS, dtype=torch.uint8, device="cuda")
        d_iq = torch.empty(nf * FIELD_SYMBOLS, dtype=torch.complex64, device="cuda")
        ranks[r].enc.encode_finish(carry.data_ptr(), d_sym.data_ptr(), d_iq.data_ptr(), stream)
        torch.cuda.synchronize()
        syms.append(d_sym.cpu().numpy())
        iqs.append(d_iq.cpu().numpy())
        carry = carry ^ pars[r]
    assert np.array_equal(np.concatenate(syms), sym_c)
    assert np.array_equal(np.concatenate(iqs).view(np.uint32), iq_c.view(np.uint32))


# ---- SDR sink adapter: fixed-point sample formats fused into the FIR store, radio-MTU chunking

def _quantize(iq_f32, full):
    """The documented formula (include/atsc_hip.h): rint(clamp(y,-1,1) * full), ties to even, exact product."""
    return np.rint(np.clip(iq_f32.astype(np.float64), -1.0, 1.0) * full)


@pytest.mark.parametrize("C_,nf", [(3, 2), (11, 1)])
@pytest.mark.parametrize("fmt,dt,full", [(atsc_amd.IQ_CS16, np.int16, 32767.0), (atsc_amd.IQ_CS8, np.int8, 127.0)])
def test_sdr_sample_formats(torch_cuda, fmt, dt, full, C_, nf):
    """CS16 / CS8 device output == the formula applied to the CF32 output of the same encode, bit for bit; and within
    one LSB (+ the 1e-5 float tolerance) of the oracle's IQ.  3 channels so that a full pair and an odd pair (dummy
    stream into the trash rows) are covered; 2 fields so that the short last FFT block of a field is followed by data; 11
    channels (AUTO still picks the FFT form) for a full group of 8 plus a partly filled one."""
    torch = torch_cuda
    per = [ts_mix(300 + c, nf) for c in range(C_)]
    ts = np.concatenate(per)
    _, iq = run_device(torch, atsc_amd.Encoder(channels=C_, max_fields=nf), ts, C_, nf)
    enc = atsc_amd.Encoder(channels=C_, max_fields=nf)
    enc.set_iq_format(fmt)
    n = C_ * nf * FIELD_SYMBOLS
    d_ts = torch.from_numpy(ts).cuda()
    guard = 4096
    d_q = torch.full((2 * n * np.dtype(dt).itemsize + guard,), 0x5A, dtype=torch.uint8, device="cuda")
    enc.encode_batch(d_ts.data_ptr(), nf, None, d_q.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    raw = d_q.cpu().numpy()
    assert (raw[-guard:] == 0x5A).all(), "store past the end of the fixed-point buffer"
    q = raw[:-guard].view(dt).astype(np.float64)
    want = _quantize(iq.view(np.float32), full)
    bad = np.flatnonzero(q != want)
    assert bad.size == 0, f"{bad.size} samples differ from the formula, first {bad[:4]}: {q[bad[:4]]} vs {want[bad[:4]]}"
    for c in range(C_):
        ref = Oracle().process_fields(per[c], nf, want=("iq",))["iq"]
        got = q.reshape(C_, -1)[c]
        assert np.abs(got - ref.astype(np.float64) * full).max() <= 0.5 + IQ_TOL * full + 1e-9


def test_sdr_format_saturates(torch_cuda):
    """Symbols outside the 8VSB alphabet (level index 255 -> 504.25) drive the FIR far beyond full scale: the fixed-point
    formats clamp to +-32767 / +-127 instead of wrapping."""
    torch = torch_cuda
    sym = torch.full((FIELD_SYMBOLS,), 255, dtype=torch.uint8, device="cuda")
    f32 = torch.empty(FIELD_SYMBOLS, dtype=torch.complex64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    atsc_amd.Encoder(channels=1, max_fields=1).mix_fir(sym.data_ptr(), 1, f32.data_ptr(), st)
    torch.cuda.synchronize()
    y = f32.cpu().numpy().view(np.float32)
    assert np.abs(y).max() > 2.0
    for fmt, dt, full in ((atsc_amd.IQ_CS16, np.int16, 32767.0), (atsc_amd.IQ_CS8, np.int8, 127.0)):
        enc = atsc_amd.Encoder(channels=1, max_fields=1)
        enc.set_iq_format(fmt)
        q = torch.empty(2 * FIELD_SYMBOLS * np.dtype(dt).itemsize, dtype=torch.uint8, device="cuda")
        enc.mix_fir(sym.data_ptr(), 1, q.data_ptr(), st)
        torch.cuda.synchronize()
        assert np.array_equal(q.cpu().numpy().view(dt).astype(np.float64), _quantize(y, full))


def test_process_chunked_radio_mtu(torch_cuda):
    """atsc_hip_process_chunked: the transmitter loop of util/atsc_encoder.cpp:84-100 -- pieces of min(remaining, MTU)
    samples, never spanning a field -- in CS16; concatenated pieces == the batch output."""
    nf, mtu = 3, 65536
    ts = ts_mt(11, nf)
    enc = atsc_amd.Encoder(channels=1, max_fields=2)
    enc.set_iq_format(atsc_amd.IQ_CS16)
    sizes, parts = [], []

    def on_chunk(ptr, nsamp, fmt):
        assert fmt == atsc_amd.IQ_CS16
        sizes.append(nsamp)
        parts.append(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int16)), shape=(2 * nsamp,)).copy())

    enc.process_chunked(ts, nf * 312, mtu, on_chunk)
    per_field = [mtu] * (FIELD_SYMBOLS // mtu) + [FIELD_SYMBOLS % mtu]
    assert sizes == per_field * nf
    ref = atsc_amd.Encoder(channels=1, max_fields=nf)
    _, iq = run_device(torch_cuda, ref, ts, 1, nf)
    assert np.array_equal(np.concatenate(parts).astype(np.float64), _quantize(iq.view(np.float32), 32767.0))
    with pytest.raises(atsc_amd.AtscHipError):
        enc.process_chunked(ts, 1, 0, on_chunk)
    with pytest.raises(atsc_amd.AtscHipError):
        enc.set_iq_format(7)


def test_null_driver_with_ingest(torch_cuda, tmp_path):
    """The null-sink driver (counterpart of util/atsc_encoder_null.cpp) end to end: junk ahead of the stream is
    skipped by the ingest front-end, whole fields are encoded, the trailing half field is dropped like the reference
    does -- or completed with null packets with `pad`; CS16 halves the bytes handed to the sink."""
    import json
    import os
    import subprocess
    exe = os.path.join(atsc_amd.ROOT, "atsc_amd", "bin", "atsc_encoder_null")
    assert os.path.exists(exe), "driver not built (atsc_amd/csrc/Makefile)"
    ts = ts_mt(8, 4)[:int(3.5 * 312) * 188]
    path = tmp_path / "in.ts"
    path.write_bytes(bytes([1, 2, 3] * 100) + ts.tobytes())
    for extra, fields, per_field, stuffed in (([], 3, FIELD_SYMBOLS * 8, 0), (["0", "cs16", "pad"], 4, FIELD_SYMBOLS * 4, 156)):
        out = subprocess.run([exe, str(path), "2"] + extra, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        r = json.loads(out.stdout)
        assert r["fields"] == fields and r["iq_bytes"] == fields * per_field
        assert r["bytes_dropped"] == 300 and r["nulls_stuffed"] == stuffed and r["sync_losses"] == 0


def test_edge_sizes(torch_cuda):
    """Ragged / extreme shapes: an empty process() call, one field, more than 64 fields in one batch (the carry scan
    walks fields in passes of 64), a field count that is not a multiple of anything."""
    enc = atsc_amd.Encoder(channels=1, max_fields=1)
    enc.process(np.zeros(0, np.uint8), 0, lambda *a: pytest.fail("no field was completed"))
    ts1 = ts_mix(1, 1)
    sym, iq = run_device(torch_cuda, enc, ts1, 1, 1)
    want = Oracle().process_fields(ts1, 1, want=("sym", "iq"))
    assert np.array_equal(sym, want["sym"]) and np.abs(iq.view(np.float32) - want["iq"]).max() <= IQ_TOL
    nf = 131
    ts = ts_mt(131, nf)
    big = atsc_amd.Encoder(channels=1, max_fields=nf)
    sym, _ = run_device(torch_cuda, big, ts, 1, nf, want_iq=False)
    assert np.array_equal(sym, Oracle().symbols(ts, nf))


# ---- matrix-core form of the mixer + FIR (fir_fused.hip, unfused instance): handles with >= 16 channels

@pytest.mark.parametrize("C_", [16, 33, 40, 64, 70])
def test_mfma_form_every_channel_against_oracle(torch_cuda, C_):
    """Channel counts around the 32-channel MFMA tile (one tile, one + a sliver, two, two + a sliver) x 3 fields (first
    field with zero history, the look-ahead into the sync-head copy at every field end): every channel's IQ against
    the oracle, symbols bit-exact, and nothing written past the buffers."""
    torch = torch_cuda
    nf = 3
    per = [ts_mix(500 + c, nf) for c in range(C_)]
    ts = np.concatenate(per)
    enc = atsc_amd.Encoder(channels=C_, max_fields=nf)
    enc.set_fir_form(atsc_amd.FIR_MFMA)
    n = C_ * nf * FIELD_SYMBOLS
    d_ts = torch.from_numpy(ts).cuda()
    d_sym = torch.empty(n, dtype=torch.uint8, device="cuda")
    guard = 1 << 16
    d_iq = torch.full((2 * n + guard,), float("nan"), dtype=torch.float32, device="cuda")
    enc.encode_batch(d_ts.data_ptr(), nf, d_sym.data_ptr(), d_iq.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    iq = d_iq.cpu().numpy()
    assert np.isnan(iq[2 * n:]).all(), "store past the end of the IQ buffer"
    iq = iq[:2 * n].reshape(C_, -1)
    assert not np.isnan(iq).any(), "samples left unwritten"
    sym = d_sym.cpu().numpy().reshape(C_, -1)
    worst = 0.0
    o = Oracle()
    for c in range(C_):
        want = Oracle().process_fields(per[c], nf, want=("sym", "iq"))
        assert np.array_equal(sym[c], want["sym"]), f"channel {c}"
        worst = max(worst, float(np.abs(iq[c] - want["iq"]).max()))
    print(f"{C_} channels: IQ max abs err {worst:.3e}")
    assert worst <= IQ_TOL
    del o
    # the two forms of the FIR agree far inside the tolerance (they are not bit-identical)
    fft = atsc_amd.Encoder(channels=C_, max_fields=nf)
    fft.set_fir_form(atsc_amd.FIR_FFT)
    _, iq_fft = run_device(torch, fft, ts, C_, nf)
    assert np.abs(iq_fft.view(np.float32).reshape(C_, -1) - iq).max() <= 2e-6
    with pytest.raises(atsc_amd.AtscHipError):
        fft.set_fir_form(5)


@pytest.mark.parametrize("C_,nf", [(33, 1), (64, 3)])
def test_mfma_form_sample_formats(torch_cuda, C_, nf):
    """CS16 / CS8 out of the fused matrix-core form (which gathers two / four tiles into whole 128-byte lines before it
    stores them) == the formula applied to the CF32 output of the unfused matrix-core form, bit for bit."""
    torch = torch_cuda
    ts = np.concatenate([ts_mt(900 + c, nf) for c in range(C_)])
    ref = atsc_amd.Encoder(channels=C_, max_fields=nf)
    ref.set_fir_form(atsc_amd.FIR_MFMA)
    _, iq = run_device(torch, ref, ts, C_, nf)
    d_ts = torch.from_numpy(ts).cuda()
    n = C_ * nf * FIELD_SYMBOLS
    for fmt, dt, full in ((atsc_amd.IQ_CS16, np.int16, 32767.0), (atsc_amd.IQ_CS8, np.int8, 127.0)):
        enc = atsc_amd.Encoder(channels=C_, max_fields=nf)
        enc.set_fir_form(atsc_amd.FIR_FUSED)
        enc.set_iq_format(fmt)
        d_q = torch.full((2 * n * np.dtype(dt).itemsize + 4096,), 0x5A, dtype=torch.uint8, device="cuda")
        enc.encode_batch(d_ts.data_ptr(), nf, None, d_q.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        raw = d_q.cpu().numpy()
        assert (raw[-4096:] == 0x5A).all()
        assert np.array_equal(raw[:-4096].view(dt).astype(np.float64), _quantize(iq.view(np.float32), full))


@pytest.mark.parametrize("C_,nf", [(16, 2), (33, 3), (64, 3), (70, 2)])
def test_fused_form_bit_identical_to_unfused(torch_cuda, C_, nf):
    """The fused form regenerates the symbols inside the FIR kernel.  Its symbol tap must equal k_trellis_map's output
    (and the oracle) bit for bit, and its IQ must equal the unfused matrix-core form's IQ bit for bit (same arithmetic
    on the same symbols); two consecutive batches exercise the carried state (reserved symbols, carries, field parity)."""
    torch = torch_cuda
    per = [ts_mix(700 + c, 2 * nf) for c in range(C_)]
    st = torch.cuda.current_stream().cuda_stream
    outs = {}
    for form in (atsc_amd.FIR_MFMA, atsc_amd.FIR_FUSED):
        enc = atsc_amd.Encoder(channels=C_, max_fields=nf)
        enc.set_fir_form(form)
        n = C_ * nf * FIELD_SYMBOLS
        res = []
        for b in range(2):
            ts = np.concatenate([p[b * nf * FIELD_TS_BYTES:(b + 1) * nf * FIELD_TS_BYTES] for p in per])
            d_ts = torch.from_numpy(ts).cuda()
            d_sym = torch.full((n + 4096,), 0xEE, dtype=torch.uint8, device="cuda")
            d_iq = torch.full((2 * n + 4096,), float("nan"), dtype=torch.float32, device="cuda")
            enc.encode_batch(d_ts.data_ptr(), nf, d_sym.data_ptr(), d_iq.data_ptr(), st)
            torch.cuda.synchronize()
            sym, iq = d_sym.cpu().numpy(), d_iq.cpu().numpy()
            assert (sym[n:] == 0xEE).all() and np.isnan(iq[2 * n:]).all(), "wrote past the buffers"
            res.append((sym[:n].reshape(C_, nf, -1), iq[:2 * n].reshape(C_, nf, -1)))
        outs[form] = res
    for b in range(2):
        s_ref, q_ref = outs[atsc_amd.FIR_MFMA][b]
        s_fus, q_fus = outs[atsc_amd.FIR_FUSED][b]
        bad = np.argwhere(s_ref != s_fus)
        assert bad.size == 0, f"batch {b}: {len(bad)} symbol mismatches, first (ch, field, pos) {bad[:5].tolist()}"
        assert not np.isnan(q_fus).any()
        assert np.array_equal(q_ref.view(np.uint32), q_fus.view(np.uint32)), f"batch {b}: IQ bits differ"
    want = Oracle().process_fields(per[C_ - 1], 2 * nf, want=("sym", "iq"))
    got_sym = np.concatenate([outs[atsc_amd.FIR_FUSED][b][0][C_ - 1].reshape(-1) for b in range(2)])
    got_iq = np.concatenate([outs[atsc_amd.FIR_FUSED][b][1][C_ - 1].reshape(-1) for b in range(2)])
    assert np.array_equal(got_sym, want["sym"])
    assert np.abs(got_iq - want["iq"]).max() <= IQ_TOL


@pytest.mark.parametrize("C_,nf", [(64, 64), (64, 16), (32, 64)])
def test_fused_bench_path_full_size(torch_cuda, C_, nf):
    """The exact path bench.py measures (BASELINE config 4 per-GPU shard: 64 distinct streams x 64 fields, IQ only ->
    fused kernel with the longest runs, 1024 tiles) and two more sizes that make the launcher pick other run lengths
    (512 and 256 tiles).  Checked at full size: the fused kernel's symbol tap equals the separate trellis kernel's
    symbols bit for bit on every channel; its IQ does not depend on whether the tap is taken (bit-identical); it agrees
    with the FFT form of the separate path within the tolerance everywhere; and the first, a middle and the last channel
    are checked against the oracle on their first and last fields."""
    torch = torch_cuda
    g = torch.Generator(device="cuda")
    g.manual_seed(77 + C_ + nf)
    ts = torch.randint(0, 256, (C_, nf * 312, 188), dtype=torch.uint8, device="cuda", generator=g)
    ts[:, :, 0] = 0x47
    st = torch.cuda.current_stream().cuda_stream
    n = nf * FIELD_SYMBOLS

    def run(form, tap):
        enc = atsc_amd.Encoder(channels=C_, max_fields=nf)
        enc.set_fir_form(form)
        d_sym = torch.empty(C_ * n, dtype=torch.uint8, device="cuda") if tap else None
        d_iq = torch.empty(C_ * n, dtype=torch.complex64, device="cuda")
        enc.encode_batch(ts.data_ptr(), nf, d_sym.data_ptr() if tap else None, d_iq.data_ptr(), st)
        torch.cuda.synchronize()
        return d_sym, torch.view_as_real(d_iq).view(C_, 2 * n)

    _, iq_bench = run(atsc_amd.FIR_AUTO, False)          # what bench.py runs
    sym_f, iq_f = run(atsc_amd.FIR_FUSED, True)
    assert bool((iq_bench.view(torch.int32) == iq_f.view(torch.int32)).all()), "the symbol tap changes the IQ"
    del iq_f
    sym_u, iq_u = run(atsc_amd.FIR_FFT, True)
    assert bool((sym_f == sym_u).all()), "fused symbol tap differs from k_trellis_map"
    assert float((iq_bench - iq_u).abs().max()) <= IQ_TOL, "fused and FFT forms disagree"
    del iq_u, sym_f
    worst = 0.0
    for c in (0, C_ // 2 + 1, C_ - 1):
        want = Oracle().process_fields(ts[c].cpu().numpy().reshape(-1), nf, want=("sym", "iq"))
        assert np.array_equal(sym_u.view(C_, n)[c].cpu().numpy(), want["sym"]), f"channel {c}"
        err = float(np.abs(iq_bench[c].cpu().numpy() - want["iq"]).max())      # every field of the channel
        assert err <= IQ_TOL, f"channel {c}: {err}"
        worst = max(worst, err)
    if (C_, nf) == (64, 64):
        # the figure bench.py quotes beside its roofline (copied to profiles/ from gpurun_out/ by hand, with the commit)
        import json
        import os
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        if os.path.isdir(out) and os.access(out, os.W_OK):
            json.dump({"max_abs_error": worst, "tolerance": IQ_TOL, "full_scale": 1.0,
                       "checked": "channels 0, 33 and 63 of 64, all 64 fields each, fused form against the oracle; every channel "
                                  "against the FFT form within the same tolerance (tests/test_gpu_parity.py::"
                                  "test_fused_bench_path_full_size)"},
                      open(os.path.join(out, "fused_iq_error.json"), "w"), indent=1)


def test_fused_randomised_cross_check(torch_cuda):
    """scripts/fuzz_fused.py, a short run: random channel counts (16..80), field counts and batch sequences; fused kernel ==
    separate kernels bit for bit (symbols, complex64 IQ), CS16 / CS8 == the formula."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_fused.py"), "8", "5"], capture_output=True,
                       text=True, timeout=280)
    assert r.returncode == 0 and "all ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
