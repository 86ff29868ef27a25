"""The interpolating form of the mixer + FIR (include/atsc_hip.h: atsc_hip_set_upsample).

    y2[2 i]     = sum_k taps[k]      z[i + k]     -- the reference's sample: bit for bit the 1x output, which the oracle pins
    y2[2 i + 1] = sum_k taps_half[k] z[i + k]     -- PARITY UNPINNED (the reference runs the radio at the symbol rate,
                                                     util/atsc_encoder.cpp:70): checked against a float64 model of this formula
                                                     built from the library's tables and the oracle's symbols, and by the spectrum.
All GPU tests call the C-ABI."""
import ctypes as C

import numpy as np
import pytest

import atsc_amd
from atsc_amd import FIELD_SYMBOLS
from oracle_lib import Oracle, ts_mix

pytestmark = pytest.mark.gpu
IQ_TOL = 1e-5
PAD = FIELD_SYMBOLS + 832


def osc_tables(nf):
    """T_f for f < nf by the reference's recursion (offset.h:35-41): re = fma(cr, er, -(ci ei)), im = fma(ci, er, cr ei) in fp32
    (emulated through float64: the products are exact there, the double rounding is far below the tolerance)."""
    t = atsc_amd.get_table(1).astype(np.float32).reshape(PAD, 2)
    er, ei = (np.float32(v) for v in atsc_amd.get_table(2))
    out = []
    cr, ci = t[:, 0].copy(), t[:, 1].copy()
    for _ in range(nf):
        out.append((cr.astype(np.float64) + 1j * ci.astype(np.float64)))
        p = (ci.astype(np.float64) * np.float64(ei)).astype(np.float32)
        q = (cr.astype(np.float64) * np.float64(ei)).astype(np.float32)
        ncr = (cr.astype(np.float64) * np.float64(er) - p.astype(np.float64)).astype(np.float32)
        nci = (ci.astype(np.float64) * np.float64(er) + q.astype(np.float64)).astype(np.float32)
        cr, ci = ncr, nci
    return out


def model(sym, nf, taps):
    """float64: sum_k taps[k] z[i + k] over every field, z = (2 v - 5.75) * T_f on the field padded with its own first
    segment (field_sync.h:16), the mixer product rounded to fp32 per component as in offset.h:24-31."""
    tabs = osc_tables(nf)
    out = np.empty(nf * FIELD_SYMBOLS, np.complex128)
    for f in range(nf):
        s = sym[f * FIELD_SYMBOLS:(f + 1) * FIELD_SYMBOLS].astype(np.float64)
        lv = 2.0 * np.concatenate([s, s[:832]]) - 5.75
        z = lv * tabs[f]
        z = z.real.astype(np.float32).astype(np.float64) + 1j * z.imag.astype(np.float32).astype(np.float64)
        out[f * FIELD_SYMBOLS:(f + 1) * FIELD_SYMBOLS] = np.correlate(z, taps.astype(np.float64), "valid")[:FIELD_SYMBOLS]
    return out


def encode(torch, ts, C_, nf, up, form=atsc_amd.FIR_AUTO, fmt=atsc_amd.IQ_CF32, tap=False):
    enc = atsc_amd.Encoder(channels=C_, max_fields=nf)
    enc.set_fir_form(form)
    enc.set_iq_format(fmt)
    enc.set_upsample(up)
    ss = {atsc_amd.IQ_CF32: 8, atsc_amd.IQ_CS16: 4, atsc_amd.IQ_CS8: 2}[fmt]
    n = C_ * nf * FIELD_SYMBOLS
    guard = 4096
    d_ts = torch.from_numpy(np.ascontiguousarray(ts)).cuda()
    d_iq = torch.full((n * up * ss + guard,), 0x5A, dtype=torch.uint8, device="cuda")
    d_sym = torch.empty(n, dtype=torch.uint8, device="cuda") if tap else None
    enc.encode_batch(d_ts.data_ptr(), nf, d_sym.data_ptr() if tap else None, d_iq.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    raw = d_iq.cpu().numpy()
    assert (raw[-guard:] == 0x5A).all(), "store past the end of the IQ buffer"
    return raw[:-guard], (d_sym.cpu().numpy() if tap else None), enc


@pytest.mark.parametrize("C_,nf", [(16, 2), (5, 1), (33, 3)])
def test_even_samples_are_the_reference_output_and_odd_samples_follow_the_formula(C_, nf):
    import torch
    per = [ts_mix(900 + c, nf) for c in range(C_)]
    ts = np.concatenate(per)
    y1, _, _ = encode(torch, ts, C_, nf, 1, form=atsc_amd.FIR_FUSED)
    y2, sym, enc = encode(torch, ts, C_, nf, 2, tap=True)
    assert enc.fir_form() == atsc_amd.FIR_FUSED and enc.upsample() == 2
    y1 = y1.view(np.complex64).reshape(C_, nf * FIELD_SYMBOLS)
    y2 = y2.view(np.complex64).reshape(C_, 2 * nf * FIELD_SYMBOLS)
    assert np.array_equal(np.ascontiguousarray(y2[:, 0::2]).view(np.uint32), y1.view(np.uint32)), "even samples differ from the 1x output"
    taps, half = atsc_amd.get_table(0), atsc_amd.get_table(7)
    assert half.shape == (112,) and np.allclose(half, half[::-1], rtol=0, atol=1e-9)      # symmetric about k = 55.5
    assert abs(half.sum() / taps.sum() - 1.0) < 2e-3                                      # same pulse, same scale
    sym = sym.reshape(C_, nf * FIELD_SYMBOLS)
    for c in sorted({0, C_ // 2, C_ - 1}):
        want = Oracle().process_fields(per[c], nf, want=("sym", "iq"))
        assert np.array_equal(sym[c], want["sym"])
        ref = want["iq"].view(np.complex64)
        assert np.abs(y2[c, 0::2] - ref).max() <= IQ_TOL                                  # the pinned half
        m0 = model(want["sym"], nf, taps[:111])
        assert np.abs(m0 - ref).max() <= 2e-6, "the float64 model does not reproduce the reference on the even samples"
        m1 = model(want["sym"], nf, half)
        err = np.abs(y2[c, 1::2] - m1).max()
        assert err <= IQ_TOL, f"channel {c}: odd samples off by {err}"


def test_unfused_instance_formats_and_spectrum():
    """atsc_hip_mix_fir on a 2x handle (caller's symbols -> the matrix-core kernel without the symbol producer) gives the fused
    kernel's bits; CS16 / CS8 are the documented formula applied to the complex64 2x output; the 2x output is band limited:
    the image of the channel around the old Nyquist frequency is far below the channel."""
    import torch
    C_, nf = 16, 2
    ts = np.concatenate([ts_mix(40 + c, nf) for c in range(C_)])
    y2, sym, _ = encode(torch, ts, C_, nf, 2, tap=True)
    n = C_ * nf * FIELD_SYMBOLS
    enc = atsc_amd.Encoder(channels=C_, max_fields=nf)
    enc.set_upsample(2)
    d_sym = torch.from_numpy(sym).cuda()
    d_iq = torch.empty(2 * n, dtype=torch.complex64, device="cuda")
    enc.mix_fir(d_sym.data_ptr(), nf, d_iq.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_iq.cpu().numpy().view(np.uint32), y2.view(np.uint32))
    yf = y2.view(np.float32).astype(np.float64)
    for fmt, dt, full in ((atsc_amd.IQ_CS16, np.int16, 32767.0), (atsc_amd.IQ_CS8, np.int8, 127.0)):
        q, _, _ = encode(torch, ts, C_, nf, 2, fmt=fmt)
        assert np.array_equal(q.view(dt).astype(np.float64), np.rint(np.clip(yf, -1.0, 1.0) * full))
    y = y2.view(np.complex64).reshape(C_, -1)[3]
    seg = y[8192:8192 + (1 << 18)].astype(np.complex128) * np.hanning(1 << 18)
    p = np.abs(np.fft.fft(seg)) ** 2
    f = np.fft.fftfreq(1 << 18)                       # cycles per output sample: the channel sits in |f| < 0.14
    inband = p[np.abs(f) < 0.13].mean()
    image = p[np.abs(f) > 0.37].mean()                # 0.5 - 0.13 and beyond
    assert 10 * np.log10(inband / image) > 40.0


def test_limits_and_streaming():
    """A 2x handle refuses the FFT form and sizes beyond one buffer descriptor; the streaming API hands out 2 x 260416 samples
    per field, equal to the batch API's."""
    import torch
    e = atsc_amd.Encoder(channels=16, max_fields=65)
    with pytest.raises(atsc_amd.AtscHipError) as err:
        e.set_upsample(2)
    assert err.value.code == atsc_amd.ERR_TOO_LARGE
    e.set_iq_format(atsc_amd.IQ_CS16)
    e.set_upsample(2)
    with pytest.raises(atsc_amd.AtscHipError) as err:
        e.set_fir_form(atsc_amd.FIR_FFT)
    assert err.value.code == atsc_amd.ERR_ARG
    with pytest.raises(atsc_amd.AtscHipError) as err:
        e.set_iq_format(atsc_amd.IQ_CF32)
    assert err.value.code == atsc_amd.ERR_TOO_LARGE
    e.close()
    e = atsc_amd.Encoder(channels=1, max_fields=2)
    e.set_fir_form(atsc_amd.FIR_FFT)
    with pytest.raises(atsc_amd.AtscHipError) as err:
        e.set_upsample(2)
    assert err.value.code == atsc_amd.ERR_ARG
    e.close()

    nf = 3
    ts = ts_mix(77, nf)
    want, _, _ = encode(torch, ts, 1, nf, 2)
    enc = atsc_amd.Encoder(channels=1, max_fields=2)
    enc.set_upsample(2)
    got = []

    def on_field(ptr, nbytes):
        got.append(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nbytes,)).copy())

    enc.process(ts, nf * 312, on_field)
    assert len(got) == nf and all(g.size == 2 * FIELD_SYMBOLS * 8 for g in got)
    assert np.array_equal(np.concatenate(got), want)


def test_time_split_spans_join_bit_identically_at_two_samples_per_symbol():
    """BASELINE config 5 on a 2x handle: 3 spans x 2 fields of 16 streams, each span a cold handle primed from a one-field
    halo with the trellis carry handed over, give the bits of one handle encoding the 6 fields in one go (a field is
    self-contained in the interpolating form as well: nothing of the filter crosses a field boundary)."""
    import torch
    from atsc_amd.timesplit import TimeSplitEncoder
    C_, R, nf = 16, 3, 2
    g = torch.Generator(device="cuda")
    g.manual_seed(321)
    ts = torch.randint(0, 256, (C_, R * nf, 312 * 188), dtype=torch.uint8, device="cuda", generator=g)
    ts.view(C_, R * nf, 312, 188)[..., 0] = 0x47
    st = torch.cuda.current_stream().cuda_stream
    cont = atsc_amd.Encoder(channels=C_, max_fields=R * nf)
    cont.set_upsample(2)
    iq_c = torch.empty(C_, R * nf, 2 * FIELD_SYMBOLS, dtype=torch.complex64, device="cuda")
    cont.encode_batch(ts.data_ptr(), R * nf, None, iq_c.data_ptr(), st)
    carry = torch.zeros(4 * C_, dtype=torch.int32, device="cuda")
    for r in range(R):
        t = TimeSplitEncoder(nf, channels=C_, upsample=2)
        t.prepare(r * nf, ts[:, r * nf - 1].contiguous() if r else None)
        span = ts[:, r * nf:(r + 1) * nf].contiguous()
        d_iq = torch.empty(C_, nf, 2 * FIELD_SYMBOLS, dtype=torch.complex64, device="cuda")
        this_carry = carry.clone()
        par = t.encode(span, None, d_iq, exchange=lambda p, grp: this_carry)
        torch.cuda.synchronize()
        same = torch.view_as_real(d_iq).view(torch.int32) == torch.view_as_real(iq_c[:, r * nf:(r + 1) * nf].contiguous()).view(torch.int32)
        assert bool(same.all()), f"span {r}: IQ bits differ"
        carry = carry ^ par
