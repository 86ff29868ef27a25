"""SDR-rate polyphase resampler (include/atsc_hip_resample.h).  PARITY UNPINNED: the reference has no
such stage (it runs the radio at the symbol rate, util/atsc_encoder.cpp:70), so the checks are against a float64 model
of the documented formula and against properties: exact pass-through of phase 0, chunking invariance, image rejection
on real modulator output.  The tap-design checks run on CPU; the kernel checks are GPU tests and call the C-ABI."""
import numpy as np
import pytest

import atsc_amd
from atsc_amd import FIELD_SYMBOLS
from oracle_lib import ts_mt

IQ_TOL = 1e-5


def model_taps(L, M, K, beta=8.0):
    n = np.arange(K * L, dtype=np.float64)
    c = (K // 2) * L
    d = n - c
    fc = 1.0 / max(L, M)
    w = np.i0(beta * np.sqrt(np.maximum(0.0, 1.0 - (d / c) ** 2))) / np.i0(beta)
    h = L * fc * np.sinc(fc * d) * w
    if M <= L:
        z = (d % L) == 0
        h[z] = 0.0
        h[c] = 1.0
    return h


def model_resample(x, L, M, K, h, n_out, q0=0):
    """y[j] = sum_k h[k*L + ph] * x[i-k] with t = q0 + j*M, i = t // L, ph = t % L; x[<0] = 0.  float64."""
    xp = np.concatenate([np.zeros(K - 1, np.complex128), x.astype(np.complex128)])
    t = q0 + np.arange(n_out, dtype=np.int64) * M
    i, ph = t // L, t % L
    y = np.zeros(n_out, np.complex128)
    for k in range(K):
        y += h[k * L + ph] * xp[i - k + K - 1]
    return y


@pytest.mark.parametrize("L,M,K", [(1, 1, 2), (2, 1, 16), (8, 7, 16), (5, 4, 24), (3, 5, 16), (125, 108, 16)])
def test_tap_design(L, M, K):
    h = atsc_amd.resampler_design(L, M, K)
    want = model_taps(L, M, K)
    assert np.abs(h - want).max() <= 2e-7
    c = (K // 2) * L
    assert np.array_equal(h[1:c], h[c + 1:][::-1][:c - 1])                  # linear phase (h[0] is the window's end point)
    if M <= L:
        assert h[c] == 1.0 and all(h[c + j * L] == 0.0 for j in range(-(K // 2), K // 2) if j)   # Nyquist, exactly
    # unit DC gain: every polyphase branch sums to ~1
    branch = h.reshape(K, L).sum(axis=0)
    assert np.abs(branch - 1.0).max() < 2e-3

def test_design_against_scipy_firwin():
    """Same prototype as scipy's Kaiser-windowed sinc (up to its DC normalisation and the forced Nyquist zeros)."""
    from scipy.signal import firwin
    L, K = 4, 16
    h = atsc_amd.resampler_design(L, 1, K).astype(np.float64)
    ref = firwin(K * L + 1, 1.0 / L, window=("kaiser", 8.0), scale=False) * L
    assert np.abs(h - ref[:K * L]).max() < 1e-6


def test_design_rejects_bad_arguments():
    lib = atsc_amd.load_library()
    buf = np.zeros(4096, np.float32)
    for L, M, K in ((0, 1, 16), (2, 5, 16), (2, 1, 15), (2, 1, 0), (257, 1, 2), (2, 1, 66)):
        assert lib.atsc_hip_resampler_design(L, M, K, 0.0, buf.ctypes.data) == -1
    import ctypes as C
    out = C.c_void_p()
    assert lib.atsc_hip_resampler_create(0, 1, 256, 1, 64, 0.0, 0, C.byref(out)) == -3      # tap table beyond the LDS budget
    assert lib.atsc_hip_resampler_create(0, 0, 2, 1, 16, 0.0, 0, C.byref(out)) == -1
    assert lib.atsc_hip_resampler_create(0, 1, 2, 1, 16, 0.0, 9, C.byref(out)) == -1


# ------------------------------------------------------------------ GPU

def _run(torch, rs, x, n_in_list):
    """x: [channels, n] complex64 numpy; resample in the given chunks; returns [channels, total_out]."""
    ch, n = x.shape
    d_x = torch.from_numpy(x).cuda()
    outs, off = [], 0
    st = torch.cuda.current_stream().cuda_stream
    esz = {atsc_amd.IQ_CF32: 8, atsc_amd.IQ_CS16: 4, atsc_amd.IQ_CS8: 2}[rs.fmt]
    for n_in in n_in_list:
        n_out = rs.out_count(n_in)
        d_y = torch.full((ch, n_out * esz + 64), 0x5A, dtype=torch.uint8, device="cuda")
        rs.resample(d_x.data_ptr() + off * 8, n_in, n, d_y.data_ptr(), n_out + 64 // esz, st)
        torch.cuda.synchronize()
        y = d_y.cpu().numpy()
        assert (y[:, n_out * esz:] == 0x5A).all(), "wrote past out_count"
        outs.append(y[:, :n_out * esz])
        off += n_in
    assert off == n
    return np.concatenate(outs, axis=1)


@pytest.mark.gpu
@pytest.mark.parametrize("L,M,K", [(1, 1, 2), (2, 1, 16), (3, 1, 16), (4, 1, 16), (2, 1, 12), (8, 7, 16), (3, 5, 16), (125, 108, 12)])
def test_gpu_matches_float64_model(L, M, K):
    import torch
    rng = np.random.default_rng(L * 100 + M)
    ch, n = 3, 50_000
    x = (rng.standard_normal((ch, n)) + 1j * rng.standard_normal((ch, n))).astype(np.complex64) * 0.3
    rs = atsc_amd.Resampler(ch, L, M, K)
    n_out = rs.out_count(n)
    assert n_out == -(-n * L // M)
    y = _run(torch, rs, x, [n]).view(np.complex64)
    h = atsc_amd.resampler_design(L, M, K).astype(np.float64)
    for c in range(ch):
        want = model_resample(x[c], L, M, K, h, n_out)
        assert np.abs(y[c] - want).max() <= 2e-6


@pytest.mark.gpu
def test_gpu_phase0_passthrough_and_chunking():
    """Integer L: y[L*i] == x[i - K/2] bit for bit; any chunking of the input gives bit-identical output.  (L = 2..4 with
    K = 16 run the sliding-window kernel k_upsample, everything else the generic k_resample: L = 2, K = 12 below is the
    generic kernel on the same data, and both are held to the same float64 model in test_gpu_matches_float64_model.)"""
    import torch
    rng = np.random.default_rng(1)
    ch, n, L, K = 2, 70_001, 2, 16
    x = (rng.standard_normal((ch, n)) + 1j * rng.standard_normal((ch, n))).astype(np.complex64) * 0.3
    whole = _run(torch, atsc_amd.Resampler(ch, L, 1, K), x, [n]).view(np.complex64)
    d = K // 2
    assert np.array_equal(np.ascontiguousarray(whole[:, L * d::L]).view(np.uint32), np.ascontiguousarray(x[:, :n - d]).view(np.uint32))
    parts = _run(torch, atsc_amd.Resampler(ch, L, 1, K), x, [1, 7, 1024, 30_000, n - 31_032]).view(np.complex64)
    assert np.array_equal(parts.view(np.uint32), whole.view(np.uint32))
    # rational ratio: the phase carried between calls keeps the output grid
    w87 = _run(torch, atsc_amd.Resampler(ch, 8, 7, K), x, [n])
    p87 = _run(torch, atsc_amd.Resampler(ch, 8, 7, K), x, [3, 4, 10_000, 5, n - 10_012])
    assert np.array_equal(w87, p87)
    # reset returns to the start of a stream
    rs = atsc_amd.Resampler(ch, 8, 7, K)
    _run(torch, rs, x[:, :1000].copy(), [1000])
    rs.reset()
    assert np.array_equal(_run(torch, rs, x, [n]), w87)


@pytest.mark.gpu
def test_gpu_modulator_output_at_twice_the_rate():
    """The real thing: 2 fields of 8VSB baseband resampled 2x (CF32 and CS16).  Spectrum: the image of the 6 MHz channel
    around the old Nyquist frequency is suppressed by > 70 dB; phase 0 is the modulator's own output."""
    import torch
    nf = 2
    ts = ts_mt(31, nf)
    enc = atsc_amd.Encoder(channels=1, max_fields=nf)
    n = nf * FIELD_SYMBOLS
    d_ts = torch.from_numpy(ts).cuda()
    d_iq = torch.empty(n, dtype=torch.complex64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    enc.encode_batch(d_ts.data_ptr(), nf, None, d_iq.data_ptr(), st)
    torch.cuda.synchronize()
    iq = d_iq.cpu().numpy().reshape(1, n)
    K = 16
    y = _run(torch, atsc_amd.Resampler(1, 2, 1, K), iq, [n]).view(np.complex64)[0]
    assert np.array_equal(np.ascontiguousarray(y[K::2]).view(np.uint32), np.ascontiguousarray(iq[0, :n - K // 2]).view(np.uint32))
    seg = y[4096:4096 + (1 << 18)].astype(np.complex128) * np.hanning(1 << 18)
    p = np.abs(np.fft.fft(seg)) ** 2
    f = np.fft.fftfreq(1 << 18)                      # cycles per output sample; the input's band is |f| < 0.25
    inband = p[np.abs(f) < 0.14].mean()              # the 8VSB channel: |f_in| < 0.28 of the symbol rate
    image = p[np.abs(f) > 0.36].mean()               # its images: 0.5 - 0.14 and beyond
    assert 10 * np.log10(inband / image) > 70.0
    q = _run(torch, atsc_amd.Resampler(1, 2, 1, K, out_format=atsc_amd.IQ_CS16), iq, [n]).view(np.int16)[0]
    assert np.array_equal(q.astype(np.float64), np.rint(np.clip(y.view(np.float32).astype(np.float64), -1, 1) * 32767))


@pytest.mark.gpu
@pytest.mark.parametrize("L", [2, 4])
def test_gpu_decimated_output_is_the_modulator_output_at_64_fields(L):
    """SURVEY 8f-1's check at BASELINE's batch length: 64 fields of two channels through the encoder and the Lx resampler on
    the device; every L-th output sample is the encoder's own IQ (whose parity with the reference the oracle pins), bit for
    bit, delayed by K/2 -- the part of the resampler's output that HAS a reference counterpart.  The other phases are
    checked against the float64 model on the first and the last 20000 samples."""
    import torch
    nf, ch, K = 64, 2, 16
    n = nf * FIELD_SYMBOLS
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    ts = torch.randint(0, 256, (ch, nf * 312, 188), dtype=torch.uint8, device="cuda", generator=g)
    ts[:, :, 0] = 0x47
    enc = atsc_amd.Encoder(channels=ch, max_fields=nf)
    d_iq = torch.empty(ch, n, dtype=torch.complex64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    enc.encode_batch(ts.data_ptr(), nf, None, d_iq.data_ptr(), st)
    rs = atsc_amd.Resampler(ch, L, 1, K)
    n_out = rs.out_count(n)
    assert n_out == L * n
    y = torch.empty(ch, n_out, dtype=torch.complex64, device="cuda")
    rs.resample(d_iq.data_ptr(), n, n, y.data_ptr(), n_out, st)
    torch.cuda.synchronize()
    d = K // 2
    same = torch.view_as_real(y[:, L * d::L]).view(torch.int32) == torch.view_as_real(d_iq[:, :n - d]).view(torch.int32)
    assert bool(same.all())
    h = model_taps(L, 1, K)
    x0 = d_iq[0].cpu().numpy()
    m = 20000
    want = model_resample(x0[:m], L, 1, K, h, L * m)
    assert np.abs(y[0, :L * m].cpu().numpy() - want).max() <= 2e-6
    i0 = n - m                                        # the last m inputs, with K - 1 samples of history in front
    want = model_resample(x0[i0 - (K - 1):], L, 1, K, h, L * (m + K - 1))[L * (K - 1):]
    assert np.abs(y[0, L * i0:].cpu().numpy() - want).max() <= 2e-6
