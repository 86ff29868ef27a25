"""Loop-back verifier (TEST INFRASTRUCTURE ONLY): 8-level symbols -> TS bytes, written from the A/53 definitions and
independent of both the oracle and the product tables.

The modulator's byte/bit stages are invertible without noise: symbol demap -> inverse precoder -> byte assembly ->
trellis de-commutation -> convolutional de-interleave -> RS syndrome check -> de-randomize.  It gives a
size-independent property for full-size batches: decode(encode(ts)) == ts and every RS(207,187)
codeword has zero syndromes.  All steps are torch gathers / XORs, so the check runs on the GPU next to the encoder
(or on the CPU for small cases).

For a signal that is NOT clean there is a maximum-likelihood decoder of the 4-state trellis code in front of that chain
(`viterbi_symbols`): the convolutional half of the code is Z0[u] = Z1[u-1] ^ Z0[u-2] with Z2 uncoded (parallel
transitions), so the decoder works on (Z1, Z0) with the nearer of the two Z2 levels as the branch metric.  It is written
as a parallel scan in the (min, +) semiring instead of a time loop, so a whole stream decodes in a few tensor operations.
`trellis_violations` is the cheap form: the number of symbols that break the code's parity check (0 for encoder output).

Conventions: data symbol j of data segment s comes from encoder (4s+j) mod 12, symbol number
t = 69s + j//12 of that encoder in the field; a byte is four (x2,x1) pairs MSB first; Z2[t] = x2[t] ^ Z2[t-1], Z1 = x1;
encoder e takes, as its g-th byte of the field, IL[12g + ((e - 4*((12g//207) % 3)) mod 12)]; IL[M] = E[M - 208*(M mod 52)].
"""
import numpy as np
import torch

SEGS, SEG_SYMS, DATA_PER_SEG = 313, 832, 828
FIELD_SYMS = SEGS * SEG_SYMS
FIELD_CODED = 312 * 207
ENC_SYMS = 312 * 828 // 12          # 21528 symbols per encoder per field
ENC_BYTES = ENC_SYMS // 4           # 5382
TAIL = 208 * 51                     # coded bytes of a field that only the next field's symbols carry


def _prbs():
    """Randomizer bytes for one field (187 per packet, restart every field): 16-bit LFSR, preset 0xF180."""
    st = 0xF180
    out = np.empty(312 * 187, np.uint8)
    for i in range(out.size):
        out[i] = ((st & 0x3C00) >> 6) | ((st & 0x40) >> 3) | ((st & 0xC) >> 1) | (st & 1)
        st <<= 1
        if st & 0x10000:
            st ^= (0x9C65 << 1) | 1
    return out


def _gf_tables():
    exp = np.zeros(512, np.int64)
    log = np.zeros(256, np.int64)
    v = 1
    for i in range(255):
        exp[i] = exp[i + 255] = v
        log[v] = i
        v = (v << 1) ^ (0x11D if v & 0x80 else 0)
    return exp, log


def _perms():
    s = np.arange(312)[:, None]
    j = np.arange(828)[None, :]
    e = (4 * s + j) % 12
    t = 69 * s + j // 12
    sym_of = np.empty((12, ENC_SYMS), np.int64)       # (encoder, symbol number) -> data-symbol index s*828+j
    sym_of[e, t] = s * 828 + j
    g = np.arange(ENC_BYTES)[None, :]
    enc = np.arange(12)[:, None]
    shift = 4 * ((12 * g // 207) % 3)
    il_of = 12 * g + (enc - shift) % 12               # (encoder, byte number) -> interleaved byte index in the field
    return sym_of, il_of


_CACHE = {}


def _consts(device):
    key = str(device)
    if key not in _CACHE:
        sym_of, il_of = _perms()
        exp, log = _gf_tables()
        _CACHE[key] = dict(sym_of=torch.from_numpy(sym_of.reshape(-1)).to(device),
                           il_of=torch.from_numpy(il_of.reshape(-1)).to(device),
                           prbs=torch.from_numpy(_prbs().reshape(312, 187)).to(device),
                           exp=torch.from_numpy(exp).to(device), log=torch.from_numpy(log).to(device))
    return _CACHE[key]


def decode_symbols(sym, fields):
    """sym: uint8 tensor [fields*260416] of level indices 0..7 of ONE stream from its first field on.
    Returns (ts_payload [(fields-1)*312, 187] uint8, worst_syndrome int): the de-randomized packet payloads (TS bytes
    1..187 of every packet) of all fields but the last (its tail lives in symbols of the next field), and the OR of all
    RS syndromes of those codewords (0 = every codeword valid)."""
    k = _consts(sym.device)
    F = fields
    v = sym.view(F, SEGS, SEG_SYMS)[:, 1:, 4:].reshape(F, 312 * 828).long()
    v = v[:, k["sym_of"]].view(F, 12, ENC_SYMS).permute(1, 0, 2).reshape(12, F * ENC_SYMS)  # per encoder, time order
    z2 = v >> 2
    x2 = z2.clone()
    x2[:, 1:] ^= z2[:, :-1]                                   # Z2[-1] = 0 at stream start
    pair = ((x2 << 1) | ((v >> 1) & 1)).view(12, F * ENC_BYTES, 4)
    byts = (pair[..., 0] << 6) | (pair[..., 1] << 4) | (pair[..., 2] << 2) | pair[..., 3]
    il = torch.empty(F, FIELD_CODED, dtype=torch.long, device=sym.device)
    il[:, k["il_of"]] = byts.view(12, F, ENC_BYTES).permute(1, 0, 2).reshape(F, 12 * ENC_BYTES)
    il = il.view(-1)
    n = torch.arange((F - 1) * FIELD_CODED, device=sym.device)
    coded = il[n + 208 * (n % 52)].view((F - 1) * 312, 207)
    # RS(207,187), generator roots alpha^0..alpha^19: S_i = sum_k c[k] * alpha^(i*(206-k)) must vanish
    lg = k["log"][coded]
    nz = coded != 0
    pw = (206 - torch.arange(207, device=sym.device))[None, :]
    worst = 0
    for i in range(20):
        term = torch.where(nz, k["exp"][(lg + i * pw) % 255], torch.zeros_like(coded))
        s = torch.zeros(coded.shape[0], dtype=torch.long, device=sym.device)
        for b in range(8):
            s |= (((term >> b) & 1).sum(dim=1) & 1) << b
        worst |= int(s.max().item())
    rows = torch.arange((F - 1) * 312, device=sym.device) % 312
    payload = (coded[:, :187] ^ k["prbs"][rows].long()).to(torch.uint8)
    return payload, worst


def _per_encoder(x, fields):
    """[fields*260416] per-symbol values of one stream -> [12, fields*21528]: the data symbols of each encoder in time order."""
    k = _consts(x.device)
    F = fields
    v = x.view(F, SEGS, SEG_SYMS)[:, 1:, 4:].reshape(F, 312 * 828)
    return v[:, k["sym_of"]].view(F, 12, ENC_SYMS).permute(1, 0, 2).reshape(12, F * ENC_SYMS)


def _from_per_encoder(v, like, fields):
    """inverse of _per_encoder: writes the data symbols back into a copy of `like` ([fields*260416])."""
    k = _consts(like.device)
    F = fields
    out = like.clone().view(F, SEGS, SEG_SYMS)
    data = torch.empty(F, 312 * 828, dtype=like.dtype, device=like.device)
    data[:, k["sym_of"]] = v.view(12, F, ENC_SYMS).permute(1, 0, 2).reshape(F, 12 * ENC_SYMS).to(like.dtype)
    out[:, 1:, 4:] = data.view(F, 312, 828)
    return out.view(-1)


def trellis_violations(sym, fields):
    """Number of data symbols of a stream (from its first field on) that break Z0[u] = Z1[u-1] ^ Z0[u-2] (zero start state)."""
    v = _per_encoder(sym.long(), fields)
    z0, z1 = v & 1, (v >> 1) & 1
    want = torch.zeros_like(z0)
    want[:, 1:] = z1[:, :-1]
    want[:, 2:] ^= z0[:, :-2]
    return int((want != z0).sum().item())


def _minplus(a, b):
    return (a.unsqueeze(-1) + b.unsqueeze(-3)).amin(dim=-2)


def _scan(m):
    """inclusive prefix products of m[..., t, 4, 4] along t in the (min, +) semiring (Hillis-Steele doubling)."""
    p = m.clone()
    T = p.shape[-3]
    d = 1
    while d < T:
        p[..., d:, :, :] = _minplus(p[..., :-d, :, :], p[..., d:, :, :])
        d *= 2
    return p


def viterbi_symbols(levels, fields):
    """Maximum-likelihood sequence decision for the data symbols of one stream received from its first field on.
    levels: float tensor [fields*260416] in level-index units (a clean symbol v arrives as float(v), v = 0..7 <-> the
    8VSB levels -7..+7).  Returns uint8 [fields*260416]: sync symbols sliced to the nearest level, data
    symbols decided by the trellis decoder (encoder state zero at the start, free at the end)."""
    dev = levels.device
    r = _per_encoder(levels.double(), fields)                   # [12, T]
    T = r.shape[1]
    # branch metric for (z1, z0): squared distance to the nearer of the two z2 levels v = 4 z2 + 2 z1 + z0
    bm = torch.empty(12, T, 2, 2, dtype=torch.float64, device=dev)
    for z1 in (0, 1):
        for z0 in (0, 1):
            lo = 2 * z1 + z0
            bm[:, :, z1, z0] = torch.minimum((r - lo) ** 2, (r - lo - 4) ** 2)
    # state s = 2 A + B, A = Z0[u], B = Z0[u-1]; input x1 (= Z1[u]): next state (x1 ^ B, A), emits (Z1, Z0) = (x1, A)
    inf = float("inf")
    M = torch.full((12, T, 4, 4), inf, dtype=torch.float64, device=dev)
    for A in (0, 1):
        for B in (0, 1):
            for x1 in (0, 1):
                M[:, :, 2 * A + B, 2 * (x1 ^ B) + A] = bm[:, :, x1, A]
    P = _scan(M)                                                # P[t] = M[0] x ... x M[t]
    alpha = torch.full((12, T + 1, 4), inf, dtype=torch.float64, device=dev)
    alpha[:, 0, 0] = 0.0                                        # both delays zero at the start of a stream
    alpha[:, 1:, :] = P[:, :, 0, :]                             # best metric from state 0 to each state after t+1 symbols
    # suffix products (reversed scan of the transposed matrices), then the best metric from each state to any end state
    S = _scan(M.flip(1).transpose(-1, -2)).flip(1).transpose(-1, -2)   # S[t] = M[t] x ... x M[T-1]
    beta = torch.zeros(12, T + 1, 4, dtype=torch.float64, device=dev)
    beta[:, :T, :] = S.amin(dim=-1)
    state = (alpha + beta).argmin(dim=-1)                       # [12, T+1]: the ML path's state before symbol t
    A, B = state >> 1, state & 1
    z0 = A[:, :T]
    z1 = A[:, 1:] ^ B[:, :T]                                    # x1 = A' ^ B
    lo = 2 * z1 + z0
    z2 = ((r - lo - 4).abs() < (r - lo).abs()).long()
    v = 4 * z2 + lo
    sliced = levels.round().clamp(0, 7).to(torch.uint8)      # segment / field syncs: nearest level (not decoded further)
    return _from_per_encoder(v.to(torch.uint8), sliced, fields)


# ---- IQ front end: complex baseband samples -> soft symbol levels.  Written from the A/53 signal definition (suppressed
# carrier VSB: real 8-level symbols + 1.25 pilot, root-raised-cosine Nyquist filtering with roll-off 0.1152, here at one
# complex sample per symbol with the carrier at phi = 2 pi (309411 - 3000000) / fs rad per sample) and NOT from
# the product's tables or the oracle: its own pulse (255 taps, not the modulator's 111), an ideal double-precision carrier,
# and a per-field data-aided estimate of the complex gain in place of the carrier / AGC loops of a real receiver (the
# reference's oscillator is not an ideal NCO: its phase walks off the ideal one by 0.4 rad over 64 fields, SURVEY.md
# "Hard parts", which is exactly what a receiver's carrier recovery absorbs).
ROLL_OFF = 0.1152
CARRIER_PHI = 2.0 * np.pi * (309411.0 - 3000000.0) / (4500000.0 / 286 * 684)
TX_DELAY = 55                      # the modulator's y[i] is centred on symbol i + 55 (111 taps)


def rrc_pulse(u, alpha=ROLL_OFF):
    """root-raised-cosine pulse, u in pulse periods (unit energy scaling left to the gain estimate)"""
    u = np.asarray(u, np.float64)
    out = np.empty_like(u)
    for idx, t in np.ndenumerate(u):
        if abs(t) < 1e-12:
            out[idx] = 1.0 - alpha + 4.0 * alpha / np.pi
        elif abs(abs(4.0 * alpha * t) - 1.0) < 1e-9:
            out[idx] = alpha / np.sqrt(2.0) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * alpha)) + (1 - 2 / np.pi) * np.cos(np.pi / (4 * alpha)))
        else:
            out[idx] = (np.sin(np.pi * t * (1 - alpha)) + 4 * alpha * t * np.cos(np.pi * t * (1 + alpha))) / (np.pi * t * (1 - (4 * alpha * t) ** 2))
    return out


def receive_iq(iq, fields, half_symbol_late=False, half_len=127):
    """iq: complex64 tensor [fields*260416], one sample per symbol, of ONE stream from its first field on (the modulator's
    output; with half_symbol_late=True the samples taken half a symbol after those, i.e. the odd samples of the 2x output).
    Returns float64 [fields*260416] soft levels in level-index units (a clean symbol v arrives as ~float(v)) for
    viterbi_symbols / decode_symbols.  Steps: matched filter (the lowpass pulse of period two samples; for the late
    stream sampled half a sample off, which puts the output back on the symbol instants), ideal carrier removed, per-field
    complex gain from the pilot then refined against the sliced symbols, real part -> level index."""
    dev = iq.device
    n = fields * FIELD_SYMS
    assert iq.numel() == n
    k = np.arange(2 * half_len + 1) - half_len + (0.5 if half_symbol_late else 0.0)
    g = rrc_pulse(k / 2.0)
    y = iq.to(torch.complex128)
    r = torch.zeros(n, dtype=torch.complex128, device=dev)                           # r[i] = sum_j g[j] y[i + j - half_len] <-> symbol i + TX_DELAY
    for j, gj in enumerate(g):                                                        # (shifted adds: no float64 convolution on every backend)
        s = j - half_len
        if s >= 0:
            r[:n - s].add_(y[s:], alpha=float(gj))
        else:
            r[-s:].add_(y[:n + s], alpha=float(gj))
    out = torch.zeros(n, dtype=torch.float64, device=dev)
    for f in range(fields):
        lo, hi = f * FIELD_SYMS, min((f + 1) * FIELD_SYMS, n - TX_DELAY)
        sym_idx = torch.arange(lo, hi, device=dev, dtype=torch.float64)
        ph = (sym_idx * CARRIER_PHI) % (2.0 * np.pi)
        d = r[lo - TX_DELAY if lo else 0:hi - TX_DELAY]
        if lo == 0:
            d = torch.cat([torch.zeros(TX_DELAY, dtype=d.dtype, device=dev), d])[:hi - lo]   # the stream's first 55 symbols: not received
        d = d * torch.polar(torch.ones_like(ph), -ph)
        a = d.mean() / 1.25                                                                # pilot: the only DC in the signal
        for _ in range(2):
            lv = (d / a).real
            dec = (((lv - 1.25 + 7.0) / 2.0).round().clamp(0, 7) * 2.0 - 7.0 + 1.25)
            a = (d * dec).sum() / (dec * dec).sum()                                        # least squares d ~ a * level
        out[lo:hi] = ((d / a).real - 1.25 + 7.0) / 2.0
    return out
