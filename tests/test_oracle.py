"""Pins the CPU restatement (oracle/atsc_oracle.c) against the reference.

(1) golden fixtures recorded from the compiled reference (tests/golden, SURVEY.md Appendix B hashes);
(2) the compiled reference itself, stage by stage, when oracle/_ref/libatsc_ref.so is present.
Byte/symbol stages: bit-exact.  IQ: <= 1e-6 abs (bit-exact on the same libm; the tolerance only
covers a different cosf/sinf on another host).
"""
import hashlib

import numpy as np
import pytest

from oracle_lib import FIELD_ENC, FIELD_SYMS, Oracle, Ref, ref_available, ts_mix, ts_mt, ts_null

IQ_TOL = 1e-6


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


CASES = {"null4": lambda: ts_null(4), "mt1234_4": lambda: ts_mt(1234, 4), "mix7_3": lambda: ts_mix(7, 3)}


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_vs_golden(golden, name):
    g, w = golden
    case = g["cases"][name]
    nf = case["fields"]
    o = Oracle()
    d = o.process_fields(CASES[name](), nf)
    assert sha(d["enc"]) == case["sha256"]["enc"]
    assert sha(Oracle.il_to_ref_layout(d["il"])) == case["sha256"]["il"]
    assert sha(d["sym"]) == case["sha256"]["sym"]
    assert d["sym"][832:848].tolist() == case["first16_data_seg0"]
    assert np.array_equal(d["sym"][:2 * 832], w[name + "_sym_head"])
    assert np.array_equal(d["sym"][FIELD_SYMS:FIELD_SYMS + 2 * 832], w[name + "_sym_f1_head"])
    assert np.array_equal(d["sym"][-832:], w[name + "_sym_tail"])
    assert np.array_equal(d["sym"][::53], w[name + "_sym_stride"])
    iq = d["iq"].view(np.complex64)
    err = np.abs(iq[w[name + "_iq_idx"]] - w[name + "_iq"]).max()
    assert err <= IQ_TOL, err
    # trellis carries in the reference's get_carry() order: z2, z0a(=D2), z0b(=D1)
    st = o.state()
    assert [st["z2"].tolist(), st["d2"].tolist(), st["d1"].tolist()] == case["carry_z2_z0a_z0b"]


def test_static_tables_vs_golden(golden):
    g, _ = golden
    assert [int(x) for x in Oracle.taps().view(np.uint32)] == g["taps_bits"]
    t0 = Oracle.osc0()
    sc = Oracle.scale()
    # libm-dependent: allow 1 ulp-ish slack in value, but expect exact bits on the build image
    ref_sc = np.array(g["osc"]["scale_bits"], np.uint32).view(np.float32)
    assert np.abs(sc - ref_sc).max() < 1e-6
    for i, bits in g["osc"]["T0_spot_bits"].items():
        v = np.array(bits, np.uint32).view(np.float32)
        assert np.abs(t0[2 * int(i):2 * int(i) + 2] - v).max() < 1e-6
    # RS generator: the well-known ATSC RS(207,187) polynomial, g[19]..g[0]
    assert Oracle.rs_gen()[::-1].tolist() == [1, 152, 185, 240, 5, 111, 99, 6, 220, 112, 150, 69, 36, 187, 22,
                                              228, 198, 121, 121, 165, 174]


def test_first_principles_byte_to_symbol():
    """A/53 byte->trellis assignment spot checks quoted in SURVEY.md 8(c): segment s symbol 0 is fed by
    trellis (4s mod 12), and a field of zeros stays level 0 except syncs (field 0, zero history)."""
    o = Oracle()
    # all-zero coded stream cannot be injected through the public path; use structure instead:
    ts = ts_null(1)
    sym = o.symbols(ts, 1)
    seg = sym.reshape(313, 832)
    assert (seg[1:, :4] == np.array([6, 1, 1, 6], np.uint8)).all()
    assert seg[0, :4].tolist() == [6, 1, 1, 6]
    assert (seg[0, 820:] == 0).all()           # first field: reserved symbols are level index 0
    assert set(np.unique(seg[0, :820])) == {1, 6}


def test_oracle_chunking_invariance():
    """Field-at-a-time == all-at-once (state carried correctly)."""
    ts = ts_mt(99, 3)
    a = Oracle().process_fields(ts, 3)
    o = Oracle()
    parts = [o.process_fields(ts[f * 312 * 188:(f + 1) * 312 * 188], 1) for f in range(3)]
    for k in ("enc", "il", "sym", "iq"):
        assert np.array_equal(a[k], np.concatenate([p[k] for p in parts]))


@pytest.mark.skipif(not ref_available(), reason="compiled reference (oracle/_ref) not built here")
@pytest.mark.parametrize("name,nf", [("mt", 6), ("mix", 5), ("null", 3)])
def test_oracle_vs_compiled_reference(name, nf):
    ts = {"mt": ts_mt(4321, nf), "mix": ts_mix(11, nf), "null": ts_null(nf)}[name]
    a = Oracle().process_fields(ts, nf)
    r = Ref()
    b = r.process_fields(ts, nf)
    assert np.array_equal(a["enc"], b["enc"])
    assert np.array_equal(Oracle.il_to_ref_layout(a["il"]), b["il"])
    assert np.array_equal(a["sym"], b["sym"])
    assert np.abs(a["iq"] - b["iq"]).max() <= IQ_TOL
    assert np.array_equal(Oracle.prbs(), Ref.prbs())
    assert np.array_equal(Oracle.taps().view(np.uint32), Ref.taps().view(np.uint32))


@pytest.mark.skipif(not ref_available(), reason="compiled reference (oracle/_ref) not built here")
def test_reference_api_chunking(golden):
    """Public process() path of the reference: arbitrary packet counts, one callback per field."""
    g, _ = golden
    ts = ts_mt(1234, 4)
    r = Ref()
    n1, iq1 = r.process(ts[:1000 * 188], 1000, 8)
    n2, iq2 = r.process(ts[1000 * 188:], 4 * 312 - 1000, 8)
    assert (n1, n2) == (3, 1)
    assert sha(np.concatenate([iq1, iq2])) == g["cases"]["mt1234_4"]["sha256"]["iq"]


def test_loopback_decoder_inverts_the_oracle():
    """tests/loopback.py (independent symbol -> TS decoder) recovers the TS payload from the oracle's symbols and finds
    every RS codeword valid; a corrupted symbol is noticed.  Pins the verifier that the full-size GPU test relies on."""
    import torch
    from loopback import decode_symbols
    nf = 3
    ts = ts_mix(99, nf)
    sym = Oracle().process_fields(ts, nf, want=("sym",))["sym"]
    payload, worst = decode_symbols(torch.from_numpy(sym.copy()), nf)
    want = ts.reshape(nf * 312, 188)[:(nf - 1) * 312, 1:]
    assert worst == 0
    assert np.array_equal(payload.numpy(), want)
    bad = sym.copy()
    bad[FIELD_SYMS + 5 * 832 + 4 + 100] ^= 4   # flip Z2 of one data symbol of field 1 (early symbols of field 0 carry pre-stream bytes)
    payload2, worst2 = decode_symbols(torch.from_numpy(bad), nf)
    assert worst2 != 0 and not np.array_equal(payload2.numpy(), want)


def test_loopback_viterbi_decodes_a_noisy_signal():
    """The verifier's trellis decoder (tests/loopback.py: viterbi_symbols, a (min,+) parallel scan over the 4-state code) on
    the oracle's symbols of two fields with additive Gaussian noise of 0.25 level steps: the plain slicer gets about four per
    cent of the data symbols wrong, the decoder recovers every one, and the chain behind it (de-interleave, RS syndromes,
    de-randomize) returns the TS payload.  Encoder output satisfies the code's parity check; the sliced noisy signal does not."""
    import torch
    from loopback import decode_symbols, trellis_violations, viterbi_symbols
    nf = 2
    ts = ts_mix(7, nf)
    sym = Oracle().process_fields(ts, nf, want=("sym",))["sym"]
    clean = torch.from_numpy(sym.copy())
    assert trellis_violations(clean, nf) == 0
    g = torch.Generator().manual_seed(11)
    rx = clean.double() + 0.25 * torch.randn(clean.shape, generator=g, dtype=torch.float64)
    sliced = rx.round().clamp(0, 7).to(torch.uint8)
    data = torch.zeros(nf, 313, 832, dtype=torch.bool)
    data[:, 1:, 4:] = True
    data = data.view(-1)
    slicer_errors = int((sliced != clean)[data].sum())
    assert slicer_errors > 0.03 * int(data.sum())
    assert trellis_violations(sliced, nf) > 1000
    dec = viterbi_symbols(rx, nf)
    assert int((dec != clean)[data].sum()) == 0
    payload, worst = decode_symbols(dec, nf)
    assert worst == 0
    assert np.array_equal(payload.numpy(), ts.reshape(nf * 312, 188)[:(nf - 1) * 312, 1:])
    # heavier noise (below the code's threshold): no longer error free, still several times better than the slicer
    rx = clean.double() + 0.5 * torch.randn(clean.shape, generator=g, dtype=torch.float64)
    dec = viterbi_symbols(rx, nf)
    sliced = rx.round().clamp(0, 7).to(torch.uint8)
    assert int((dec != clean)[data].sum()) * 3 < int((sliced != clean)[data].sum())


def test_loopback_receiver_recovers_the_ts_from_the_oracles_iq():
    """tests/loopback.py: receive_iq (own matched filter, ideal carrier, pilot gain) + slicer + decode_symbols on the oracle's
    complex64 IQ of three fields gives the TS back: the verifier's IQ front end is right before the GPU tests lean on it."""
    import torch
    from loopback import decode_symbols, receive_iq
    nf = 3
    ts = ts_mt(5, nf)
    w = Oracle().process_fields(ts, nf, want=("sym", "iq"))
    lv = receive_iq(torch.from_numpy(w["iq"].view(np.complex64).copy()), nf)
    sym = torch.from_numpy(w["sym"].astype(np.int64))
    assert float((lv - sym.double())[64:-400].abs().max()) < 0.15
    payload, worst = decode_symbols(lv.round().clamp(0, 7).to(torch.uint8), nf)
    assert worst == 0
    assert bool((payload == torch.from_numpy(ts.reshape(nf * 312, 188)[:(nf - 1) * 312, 1:].copy())).all())
