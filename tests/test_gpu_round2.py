"""GPU parity tests added in round 2: the corners of the benchmarked (fused) path -- time-split hand-off with a
non-null carry on 32- and 64-channel handles, state export / import on a fused handle, BASELINE config 5 at its
stated 8 x 64 fields -- and the argument / tap fixes of the C-ABI host layer.  Everything goes through the C-ABI."""
import ctypes as C

import numpy as np
import pytest

import atsc_amd
from atsc_amd import FIELD_SYMBOLS, FIELD_TS_BYTES
from oracle_lib import Oracle, ts_mix, ts_mt
from test_gpu_parity import IQ_TOL, d2h, run_device

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _streams(torch, C_, nf, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    ts = torch.randint(0, 256, (C_, nf * 312, 188), dtype=torch.uint8, device="cuda", generator=g)
    ts[:, :, 0] = 0x47
    return ts.view(C_, nf, FIELD_TS_BYTES)


def _encode(torch, enc, ts, C_, nf, tap=True):
    d_sym = torch.empty(C_ * nf * FIELD_SYMBOLS, dtype=torch.uint8, device="cuda") if tap else None
    d_iq = torch.empty(C_ * nf * FIELD_SYMBOLS, dtype=torch.complex64, device="cuda")
    enc.encode_batch(ts.data_ptr(), nf, d_sym.data_ptr() if tap else None, d_iq.data_ptr(),
                     torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_sym, torch.view_as_real(d_iq).view(C_, nf, 2 * FIELD_SYMBOLS)


@pytest.mark.parametrize("C_", [32, 64])
def test_config5_time_split_fused_handles(torch_cuda, C_):
    """BASELINE config 5 with handles that run the FUSED kernel (32 / 64 streams split at the same field boundaries):
    3 spans x 2 fields, each span a cold handle primed from a one-field halo, encode_finish with a NON-NULL carry ->
    symbols and IQ bit-identical to one handle encoding the 6 fields continuously; one channel against the oracle."""
    from atsc_amd.timesplit import TimeSplitEncoder
    torch = torch_cuda
    R, nf = 3, 2
    ts = _streams(torch, C_, R * nf, 900 + C_)
    cont = atsc_amd.Encoder(channels=C_, max_fields=R * nf)
    assert cont.fir_form() == atsc_amd.FIR_FUSED
    sym_c, iq_c = _encode(torch, cont, ts.contiguous(), C_, R * nf)
    sym_c = sym_c.view(C_, R * nf, FIELD_SYMBOLS)
    carry = torch.zeros(4 * C_, dtype=torch.int32, device="cuda")
    for r in range(R):
        t = TimeSplitEncoder(nf, channels=C_)
        assert t.enc.fir_form() == atsc_amd.FIR_FUSED
        t.prepare(r * nf, ts[:, r * nf - 1].contiguous() if r else None)
        span = ts[:, r * nf:(r + 1) * nf].contiguous()
        d_sym = torch.empty(C_ * nf * FIELD_SYMBOLS, dtype=torch.uint8, device="cuda")
        d_iq = torch.empty(C_ * nf * FIELD_SYMBOLS, dtype=torch.complex64, device="cuda")
        this_carry = carry.clone()
        par = t.encode(span, d_sym, d_iq, exchange=lambda p, g: this_carry)
        torch.cuda.synchronize()
        assert r == 0 or int(this_carry.abs().sum()) != 0, "the carry handed to the fused branch must not be trivial"
        assert bool((d_sym.view(C_, nf, FIELD_SYMBOLS) == sym_c[:, r * nf:(r + 1) * nf]).all()), f"span {r}: symbols differ"
        got = torch.view_as_real(d_iq).view(C_, nf, 2 * FIELD_SYMBOLS)
        assert bool((got.view(torch.int32) == iq_c[:, r * nf:(r + 1) * nf].view(torch.int32)).all()), f"span {r}: IQ bits differ"
        carry = carry ^ par
    c = C_ - 3
    want = Oracle().process_fields(ts[c].cpu().numpy().reshape(-1), R * nf, want=("sym", "iq"))
    assert np.array_equal(sym_c[c].cpu().numpy().reshape(-1), want["sym"])
    assert np.abs(iq_c[c].cpu().numpy().reshape(-1) - want["iq"]).max() <= IQ_TOL


def test_state_roundtrip_on_fused_handle(torch_cuda):
    """export_state / import_state of all 64 channels of a fused handle after 2 fields: a second handle continues
    bit-identically (symbols and IQ) to the handle that encodes the 4 fields in two batches itself."""
    torch = torch_cuda
    C_, nf = 64, 2
    ts = _streams(torch, C_, 2 * nf, 4711)
    a = atsc_amd.Encoder(channels=C_, max_fields=nf)
    assert a.fir_form() == atsc_amd.FIR_FUSED
    _encode(torch, a, ts[:, :nf].contiguous(), C_, nf)
    states = [a.export_state(c) for c in range(C_)]
    assert all(s.field_index == nf for s in states)
    sym_a, iq_a = _encode(torch, a, ts[:, nf:].contiguous(), C_, nf)
    b = atsc_amd.Encoder(channels=C_, max_fields=nf)
    for c in range(C_):
        b.import_state(states[c], c)
    sym_b, iq_b = _encode(torch, b, ts[:, nf:].contiguous(), C_, nf)
    assert bool((sym_a == sym_b).all())
    assert bool((iq_a.view(torch.int32) == iq_b.view(torch.int32)).all())
    # and the trellis state of a few channels equals the oracle's delay elements
    for c in (0, 31, 63):
        o = Oracle()
        o.symbols(ts[c, :nf].cpu().numpy().reshape(-1), nf)
        s = o.state()
        assert list(states[c].z2) == s["z2"].tolist() and list(states[c].z0_next) == s["d1"].tolist()
        assert list(states[c].z0_prev) == s["d2"].tolist()


def test_config5_at_stated_size_8x64(torch_cuda):
    """BASELINE config 5 at its stated size on one GPU: one 512-field stream as 8 spans of 64 fields (cold handles,
    one-field halos, parity hand-off) -> symbols and IQ bit-identical to the continuous 512-field encode; the first
    and last span's first field are checked against the oracle through the continuous result."""
    from atsc_amd.timesplit import TimeSplitEncoder
    torch = torch_cuda
    R, nf = 8, 64
    ts = _streams(torch, 1, R * nf, 512)
    cont = atsc_amd.Encoder(channels=1, max_fields=R * nf)
    sym_c, iq_c = _encode(torch, cont, ts.contiguous(), 1, R * nf)
    sym_c = sym_c.view(R, nf * FIELD_SYMBOLS)
    iq_c = iq_c.view(R, nf * 2 * FIELD_SYMBOLS)
    del cont
    carry = torch.zeros(4, dtype=torch.int32, device="cuda")
    d_sym = torch.empty(nf * FIELD_SYMBOLS, dtype=torch.uint8, device="cuda")
    d_iq = torch.empty(nf * FIELD_SYMBOLS, dtype=torch.complex64, device="cuda")
    for r in range(R):
        t = TimeSplitEncoder(nf)
        t.prepare(r * nf, ts[:, r * nf - 1].contiguous() if r else None)
        this_carry = carry.clone()
        par = t.encode(ts[:, r * nf:(r + 1) * nf].contiguous(), d_sym, d_iq, exchange=lambda p, g: this_carry)
        torch.cuda.synchronize()
        assert bool((d_sym == sym_c[r]).all()), f"span {r}: symbols differ"
        assert bool((torch.view_as_real(d_iq).view(-1).view(torch.int32) == iq_c[r].view(torch.int32)).all()), f"span {r}: IQ bits differ"
        carry = carry ^ par
        del t
    o = Oracle()
    want = o.process_fields(ts[0, :2].cpu().numpy().reshape(-1), 2, want=("sym", "iq"))
    assert np.array_equal(sym_c[0, :2 * FIELD_SYMBOLS].cpu().numpy(), want["sym"])
    assert np.abs(iq_c[0, :4 * FIELD_SYMBOLS].cpu().numpy() - want["iq"]).max() <= IQ_TOL


def test_encode_batch_host_on_a_fused_handle(torch_cuda):
    """atsc_hip_encode_batch_host with symbols AND IQ requested on a 32-channel handle (AUTO -> fused kernel, which keeps
    no internal symbol buffer): the symbols must be the batch's own, every channel against the oracle."""
    C_, nf = 32, 2
    per = [ts_mix(1300 + c, nf) for c in range(C_)]
    enc = atsc_amd.Encoder(channels=C_, max_fields=nf)
    assert enc.fir_form() == atsc_amd.FIR_FUSED
    sym, iq = enc.encode_batch_host(np.concatenate(per), nf)
    sym = sym.reshape(C_, -1)
    iq = iq.view(np.float32).reshape(C_, -1)
    for c in range(C_):
        want = Oracle().process_fields(per[c], nf, want=("sym", "iq"))
        assert np.array_equal(sym[c], want["sym"]), f"channel {c}"
        assert np.abs(iq[c] - want["iq"]).max() <= IQ_TOL, f"channel {c}"
    # symbols only / IQ only
    enc2 = atsc_amd.Encoder(channels=C_, max_fields=nf)
    sym2, none = enc2.encode_batch_host(np.concatenate(per), nf, want_iq=False)
    assert none is None and np.array_equal(sym2.reshape(C_, -1), sym)


def test_mix_fir_argument_checks(torch_cuda):
    """atsc_hip_mix_fir refuses a batch larger than the handle's oscillator table (ATSC_HIP_ERR_TOO_LARGE, -3) and a
    call between encode_begin and encode_finish (ATSC_HIP_ERR_STATE, -4); a failed call leaves the handle usable."""
    torch = torch_cuda
    nf = 2
    enc = atsc_amd.Encoder(channels=1, max_fields=nf)
    L = enc.L
    sym = torch.zeros(3 * FIELD_SYMBOLS, dtype=torch.uint8, device="cuda")
    iq = torch.empty(3 * FIELD_SYMBOLS, dtype=torch.complex64, device="cuda")
    assert L.atsc_hip_mix_fir(enc.h, sym.data_ptr(), 3, iq.data_ptr(), None) == -3
    ts = torch.from_numpy(ts_mt(8, nf)).cuda()
    enc.encode_begin(ts.data_ptr(), nf)
    assert L.atsc_hip_mix_fir(enc.h, sym.data_ptr(), 1, iq.data_ptr(), None) == -4
    assert L.atsc_hip_encode_batch_host(enc.h, ts.cpu().numpy().ctypes.data, nf, None, None) == -4
    d_sym = torch.empty(nf * FIELD_SYMBOLS, dtype=torch.uint8, device="cuda")
    # a misaligned pointer is refused and the batch stays pending; the retry succeeds
    assert L.atsc_hip_encode_finish(enc.h, None, d_sym.data_ptr() + 1, None, None) == -1
    enc.encode_finish(None, d_sym.data_ptr(), iq.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_sym.cpu().numpy(), Oracle().symbols(ts.cpu().numpy(), nf))


def test_trellis_taps_with_fewer_fields_than_the_handle_holds(torch_cuda):
    """TRELLIS_PAR / TRELLIS_SCAN taps on a 3-channel handle made for 4 fields after a 2-field batch: the buffers are
    packed with the batch's field count, and the reported stride must say so.  Checked through the scan's own
    invariants per channel: cin[0] = state entering the batch (zero), cin[i+1] = cin[i] ^ par[i], and the XOR of all
    parities = the span parity."""
    torch = torch_cuda
    C_, nf = 3, 2
    per = [ts_mix(40 + c, nf) for c in range(C_)]
    enc = atsc_amd.Encoder(channels=C_, max_fields=4)
    run_device(torch, enc, np.concatenate(per), C_, nf, want_iq=False)
    p_par, b_par, st_par = enc.stage_buffer(atsc_amd.STAGE_TRELLIS_PAR)
    p_cin, b_cin, st_cin = enc.stage_buffer(atsc_amd.STAGE_TRELLIS_SCAN)
    assert st_par == st_cin == nf * 26 * 16 and b_par == C_ * st_par
    par = d2h(p_par, b_par).view(np.uint32).reshape(C_, nf * 26, 4)
    cin = d2h(p_cin, b_cin).view(np.uint32).reshape(C_, nf * 26, 4)
    for c in range(C_):
        assert not cin[c, 0, :3].any()
        assert np.array_equal(cin[c, 1:, :3], cin[c, :-1, :3] ^ par[c, :-1, :3]), f"channel {c}"
        total = np.bitwise_xor.reduce(par[c, :, :3], axis=0)
        sp = enc.span_parity(c)                     # [3][12]: z2, D1, D2 per encoder
        packed = (sp[0] | (sp[1] << 1) | (sp[2] << 2)).astype(np.uint8).view(np.uint32)
        assert np.array_equal(total, packed), f"channel {c}"
        # and against the oracle's delay elements after the batch (the stream started from zero state)
        o = Oracle()
        o.symbols(per[c], nf)
        s = o.state()
        assert sp[0].tolist() == s["z2"].tolist() and sp[1].tolist() == s["d1"].tolist() and sp[2].tolist() == s["d2"].tolist()


def test_streaming_pipeline_many_batches_per_call(torch_cuda):
    """atsc_hip_process with ONE call carrying 7 fields on a handle made for 2 fields per batch: four batches run as a
    two-slot pipeline (upload + kernels of batch k+1 under the download and callbacks of batch k); the callbacks still
    come one per field, in order, before the call returns, each buffer intact while its callback runs."""
    nf = 7
    ts = ts_mt(2718, nf)
    want = Oracle().process_fields(ts, nf, want=("iq",))["iq"].reshape(nf, -1)
    enc = atsc_amd.Encoder(channels=1, max_fields=2)
    got = []

    def on_field(ptr, nbytes):
        assert nbytes == FIELD_SYMBOLS * 8
        got.append(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(FIELD_SYMBOLS * 2,)).copy())

    enc.process(ts, nf * 312 - 5, on_field)
    assert len(got) == nf - 1
    enc.process(ts[(nf * 312 - 5) * 188:], 5, on_field)
    assert len(got) == nf
    for f in range(nf):
        assert np.abs(got[f] - want[f]).max() <= IQ_TOL, f"field {f}"


def test_reference_null_driver_runs_on_the_drop_in(torch_cuda, tmp_path):
    """The reference's own util/atsc_encoder_null.cpp, compiled unmodified against the adapter header in the build
    container (oracle/Makefile -> oracle/_ref/ref_encoder_null_hip), runs a 3.5-field TS file through the GPU path and
    exits cleanly (it is a null sink: the reference's driver discards the IQ, util/atsc_encoder_null.cpp:41-42)."""
    import os
    import subprocess
    exe = os.path.join(atsc_amd.ROOT, "oracle", "_ref", "ref_encoder_null_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_encoder_null_hip was not built (needs the reference tree at build time)")
    path = tmp_path / "in.ts"
    path.write_bytes(ts_mt(8, 4)[:int(3.5 * 312) * 188].tobytes())
    out = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1000:]
    assert "Failed" not in out.stdout


@pytest.mark.gpu
def test_matrix_core_forms_respect_the_descriptor_limit():
    """16 channel rows of the largest batch's IQ must fit one 32-bit-offset buffer descriptor (include/atsc_hip.h): a
    handle of 129 complex64 fields resolves AUTO to FFT and refuses MFMA / FUSED; with CS16 samples the same handle fits."""
    e = atsc_amd.Encoder(channels=16, max_fields=129)
    assert e.fir_form() == atsc_amd.FIR_FFT
    for form in (atsc_amd.FIR_MFMA, atsc_amd.FIR_FUSED):
        with pytest.raises(atsc_amd.AtscHipError) as err:
            e.set_fir_form(form)
        assert err.value.code == atsc_amd.ERR_TOO_LARGE
    e.set_iq_format(atsc_amd.IQ_CS16)
    assert e.fir_form() == atsc_amd.FIR_FUSED
    e.set_fir_form(atsc_amd.FIR_FUSED)
    with pytest.raises(atsc_amd.AtscHipError) as err:
        e.set_iq_format(atsc_amd.IQ_CF32)
    assert err.value.code == atsc_amd.ERR_TOO_LARGE
    e.close()
    e = atsc_amd.Encoder(channels=16, max_fields=128)
    assert e.fir_form() == atsc_amd.FIR_FUSED
    e.close()


def test_seek_equals_the_export_import_route(torch_cuda):
    """atsc_hip_seek (all channels, on the device, stream-ordered) against what it replaces: per channel
    export_state -> zero the trellis state, set the field index -> import_state through the host.  After a halo field both
    handles encode the same span with the same carry: symbols and IQ bit-identical, for a target ahead of the handle's
    field index (oscillator advanced from its current state), one behind it (from T_0), the index it already has and 0."""
    torch = torch_cuda
    C_, nf = 3, 2
    st = torch.cuda.current_stream().cuda_stream
    ts = _streams(torch, C_, 1 + nf, 77)
    halo, span = ts[:, 0].contiguous(), ts[:, 1:].contiguous()
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    carry = torch.randint(0, 8, (4 * C_, 4), dtype=torch.uint8, device="cuda", generator=g)
    carry[3::4] = 0                                       # the pad word of every channel
    carry = carry.view(torch.int32).view(-1).contiguous()
    a = atsc_amd.Encoder(channels=C_, max_fields=nf)
    b = atsc_amd.Encoder(channels=C_, max_fields=nf)
    for target in (40, 3, 3, 1, 0, 1000):                 # ahead, behind, same, ..., far ahead
        outs = []
        for enc, use_seek in ((a, True), (b, False)):
            enc.encode_batch(halo.data_ptr(), 1, None, None, st)
            if use_seek:
                enc.seek(target, st)
            else:
                for c in range(C_):
                    s = enc.export_state(c)
                    for i in range(12):
                        s.z2[i] = s.z0_next[i] = s.z0_prev[i] = 0
                    s.field_index = target
                    enc.import_state(s, c)
            d_sym = torch.empty(C_ * nf * FIELD_SYMBOLS, dtype=torch.uint8, device="cuda")
            d_iq = torch.empty(C_ * nf * FIELD_SYMBOLS, dtype=torch.complex64, device="cuda")
            par = torch.zeros(4 * C_, dtype=torch.int32, device="cuda")
            enc.encode_begin(span.data_ptr(), nf, par.data_ptr(), st)
            if use_seek:
                with pytest.raises(atsc_amd.AtscHipError) as ei:
                    enc.seek(target + 1, st)              # not between begin and finish
                assert ei.value.code == atsc_amd.ERR_STATE
            enc.encode_finish(carry.data_ptr(), d_sym.data_ptr(), d_iq.data_ptr(), st)
            torch.cuda.synchronize()
            outs.append((d_sym, torch.view_as_real(d_iq).view(torch.int32), par))
        assert bool((outs[0][0] == outs[1][0]).all()), f"target {target}: symbols differ"
        assert bool((outs[0][1] == outs[1][1]).all()), f"target {target}: IQ bits differ"
        assert bool((outs[0][2] == outs[1][2]).all())
        assert a.export_state(0).field_index == target + nf
