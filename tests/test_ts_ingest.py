"""TS ingest front-end (include/atsc_ts_ingest.h) against a Python model of its specification.
Host logic only: runs on CPU.  The reference has no such stage (its callers fread aligned fields,
util/atsc_encoder_null.cpp:38-43), so the behaviour is pinned by the model below and by the invariants that a clean
stream passes through unchanged and that the result does not depend on how the input is chunked."""
import numpy as np
import pytest

import atsc_amd
from oracle_lib import ts_mt, ts_null

PKT = 188
NULL = np.array([0x47, 0x1F, 0xFF, 0x10] + [0xFF] * 184, np.uint8)


def model(raw, lock=5, loss=3):
    """Specification: returns (packets bytes, stats dict) for the whole byte string pushed at once."""
    raw = bytes(raw)
    out = bytearray()
    st = dict(bytes_dropped=0, sync_errors=0, sync_losses=0)
    pos, locked, n = 0, False, len(raw)
    while True:
        if not locked:
            need = (lock - 1) * PKT + 1
            start = pos
            found = False
            while n >= pos + need:
                if all(raw[pos + k * PKT] == 0x47 for k in range(lock)):
                    found = True
                    break
                pos += 1
            st["bytes_dropped"] += pos - start
            if not found:
                break
            locked = True
        held, bad = 0, 0
        while n - pos - held >= PKT:
            if raw[pos + held] == 0x47:
                for q in range(0, held, PKT):
                    out += b"\x47" + raw[pos + q + 1:pos + q + PKT]
                    st["sync_errors"] += 1
                out += raw[pos + held:pos + held + PKT]
                pos += held + PKT
                held, bad = 0, 0
            else:
                held += PKT
                bad += 1
                if bad >= loss:
                    locked = False
                    st["sync_losses"] += 1
                    st["bytes_dropped"] += 1
                    pos += 1
                    break
        if locked:
            break
    return np.frombuffer(bytes(out), np.uint8), st


def junk(rng, n):
    """Random bytes without 0x47, so that the model's answer is unambiguous where the test says so."""
    b = rng.integers(0, 255, n, dtype=np.uint8)
    b[b >= 0x47] += 1
    return b


def run_lib(raw, chunks, **kw):
    ing = atsc_amd.TsIngest(**kw)
    off = 0
    for c in chunks:
        ing.push(raw[off:off + c])
        off += c
    assert off == len(raw)
    got = ing.pop(ing.available() + 1)
    return got, ing.stats(), ing


def test_clean_stream_passes_through_unchanged():
    ts = ts_mt(9, 2)
    got, st, _ = run_lib(ts, [len(ts)])
    assert np.array_equal(got, ts)
    assert st["bytes_dropped"] == 0 and st["sync_errors"] == 0 and st["sync_losses"] == 0 and st["locked"] == 1
    assert st["packets_out"] == 2 * 312 and st["bytes_in"] == ts.size


@pytest.mark.parametrize("seed", range(6))
def test_matches_model_and_is_chunking_invariant(seed):
    rng = np.random.default_rng(seed)
    pk = ts_mt(100 + seed, 1).reshape(312, PKT).copy()
    parts = [junk(rng, int(rng.integers(0, 700)))]
    i = 0
    while i < 312:
        run = min(int(rng.integers(1, 60)), 312 - i)
        blk = pk[i:i + run].copy()
        kind = rng.integers(0, 4)
        if kind == 1 and run > 6:          # single sync-byte bit errors: passed through, byte repaired
            blk[rng.integers(1, run - 1), 0] ^= 0x10
        parts.append(blk.reshape(-1))
        if kind == 2:                      # a slip: bytes lost / junk inserted -> loss of lock and re-acquisition
            parts.append(junk(rng, int(rng.integers(1, 400))))
        if kind == 3:                      # truncated packet
            parts.append(pk[i, :int(rng.integers(1, 187))].copy())
        i += run
    raw = np.concatenate(parts)
    want, wst = model(raw)
    got1, st1, _ = run_lib(raw, [len(raw)])
    assert np.array_equal(got1, want)
    for k in wst:
        assert st1[k] == wst[k], k
    # byte-at-a-time and random chunking give the same packets and the same counters
    cuts = np.sort(rng.integers(0, len(raw), 50))
    chunks = np.diff(np.concatenate([[0], cuts, [len(raw)]])).tolist()
    got2, st2, _ = run_lib(raw, chunks)
    assert np.array_equal(got2, want) and {k: st2[k] for k in wst} == wst
    got3, st3, _ = run_lib(raw[:5000], [1] * 5000)
    want3, wst3 = model(raw[:5000])
    assert np.array_equal(got3, want3) and {k: st3[k] for k in wst3} == wst3
    # every output packet starts with a sync byte
    assert (got1.reshape(-1, PKT)[:, 0] == 0x47).all()


def test_payload_0x47_does_not_fool_the_hunt():
    """Payload full of 0x47 at non-188 spacing ahead of the true packets: lock lands on the true boundary."""
    rng = np.random.default_rng(3)
    pk = ts_mt(4, 1)[:40 * PKT].copy()
    pre = junk(rng, 300)
    pre[::7] = 0x47
    raw = np.concatenate([pre, pk])
    want, _ = model(raw)
    got, st, _ = run_lib(raw, [len(raw)])
    assert np.array_equal(got, want)
    assert np.array_equal(got[-10 * PKT:], pk[-10 * PKT:])


def test_null_stuffing_and_field_padding():
    ts = ts_mt(5, 1)[:100 * PKT]
    ing = atsc_amd.TsIngest()
    ing.push(ts)
    out, stuffed = ing.pop_padded(312)
    assert stuffed == 212 and out.size == 312 * PKT
    assert np.array_equal(out[:100 * PKT], ts)
    assert (out[100 * PKT:].reshape(-1, PKT) == NULL).all()
    st = ing.stats()
    assert st["nulls_stuffed"] == 212 and st["packets_out"] == 312
    # the stuffed packet is the null packet the oracle's null input uses
    assert np.array_equal(ts_null(1)[:PKT], NULL)
    # underrun of an empty source: a whole field of nulls
    out, stuffed = ing.pop_padded(312)
    assert stuffed == 312 and np.array_equal(out, ts_null(1))


def test_large_push_is_linear_time():
    import time
    ts = np.tile(ts_mt(1, 1), 200)     # 11.7 MB in one push
    ing = atsc_amd.TsIngest()
    t0 = time.perf_counter()
    n = ing.push(ts)
    assert n == 200 * 312
    assert time.perf_counter() - t0 < 2.0
    assert np.array_equal(ing.pop(n), ts)


def test_bad_arguments():
    with pytest.raises(atsc_amd.AtscHipError):
        atsc_amd.TsIngest(lock_packets=1)
    with pytest.raises(atsc_amd.AtscHipError):
        atsc_amd.TsIngest(loss_packets=-1)
    lib = atsc_amd.load_library()
    assert lib.atsc_ts_ingest_push(None, None, 0) == 0 and lib.atsc_ts_ingest_available(None) == 0
