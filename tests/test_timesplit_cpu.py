"""N>1 host path on CPU: the one exchange step of the time-split mode over gloo, world_size 2,
with span parities taken from the oracle (state_out XOR state_in of each span)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle_lib import FIELD_TS, Oracle, ts_mt


def pack(z2, d1, d2):
    """16-byte span-parity / carry format of atsc_hip_encode_begin: byte per encoder, bit0=z2, bit1=D1, bit2=D2."""
    b = (z2 | (d1 << 1) | (d2 << 2)).astype(np.uint8)
    return np.concatenate([b, np.zeros(4, np.uint8)]).view(np.int32).copy()


def oracle_states(ts, spans):
    o = Oracle()
    states = [pack(np.zeros(12, np.uint8), np.zeros(12, np.uint8), np.zeros(12, np.uint8))]
    off = 0
    for n in spans:
        o.symbols(ts[off * FIELD_TS:(off + n) * FIELD_TS], n)
        off += n
        s = o.state()
        states.append(pack(s["z2"], s["d1"], s["d2"]))
    return states


def _worker(rank, world, port, pars, expect, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import atsc_amd
    from atsc_amd.timesplit import check_uniform, exclusive_xor_over_ranks
    carry = exclusive_xor_over_ranks(torch.from_numpy(pars[rank]))
    ok = bool(np.array_equal(carry.numpy(), expect[rank]))
    # every rank must run the same channel count / FIR form / sample format, or the spans do not join bit-identically
    check_uniform((1, atsc_amd.FIR_FFT, atsc_amd.IQ_CF32))
    try:
        check_uniform((1, atsc_amd.FIR_FFT if rank == 0 else atsc_amd.FIR_FUSED, atsc_amd.IQ_CF32))
        ok = False
    except atsc_amd.AtscHipError as e:
        ok = ok and "disagree" in str(e)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_gloo_world2():
    spans = [2, 2]
    ts = ts_mt(31, sum(spans))
    st = oracle_states(ts, spans)
    pars = [st[i + 1] ^ st[i] for i in range(len(spans))]      # span parity = state_out XOR state_in (linearity)
    expect = [st[0], st[1]]                                     # state entering each span (stream starts at zero)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, pars, expect, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == {0: True, 1: True}


def test_exchange_without_process_group_is_identity():
    from atsc_amd.timesplit import exclusive_xor_over_ranks
    par = torch.tensor([1, 2, 3, 0], dtype=torch.int32)
    assert int(exclusive_xor_over_ranks(par).abs().sum()) == 0
