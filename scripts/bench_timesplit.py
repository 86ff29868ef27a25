#!/usr/bin/env python3
"""Hand-off cost of the time-split path (BASELINE config 5): one stream, this rank's span of 64 fields.
Times, per span: encode_begin (state-independent stages + span parity), the exchange (all-gather of 16 B per rank over
RCCL + local XOR fold), encode_finish.  Run under torchrun with one rank per GPU; with one rank the exchange number is the
collective's fixed cost (launch + completion), the part every rank pays before its finish stage can start.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/bench_timesplit.py"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import atsc_amd  # noqa: E402
from atsc_amd.timesplit import TimeSplitEncoder, exclusive_xor_over_ranks  # noqa: E402

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
if "RANK" in os.environ:
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
fields, reps = 64, 20
g = torch.Generator(device="cuda")
g.manual_seed(99)
ts_all = torch.randint(0, 256, ((world * fields + 1) * 312, 188), dtype=torch.uint8, device="cuda", generator=g)
ts_all[:, 0] = 0x47
first = rank * fields
tse = TimeSplitEncoder(fields, device=local)
d_ts = ts_all[first * 312:(first + fields) * 312].contiguous()
d_iq = torch.empty(fields * atsc_amd.FIELD_SYMBOLS, dtype=torch.complex64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
t_begin, t_xchg, t_finish = [], [], []
for it in range(reps + 3):
    tse.prepare(first, ts_all[(first - 1) * 312:first * 312].contiguous() if first else None)
    par = torch.zeros(4, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    tse.enc.encode_begin(d_ts.data_ptr(), fields, par.data_ptr(), stream)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    carry = exclusive_xor_over_ranks(par).contiguous()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    tse.enc.encode_finish(carry.data_ptr(), None, d_iq.data_ptr(), stream)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    if it >= 3:
        t_begin.append(t1 - t0); t_xchg.append(t2 - t1); t_finish.append(t3 - t2)
med = lambda v: sorted(v)[len(v) // 2] * 1e3
out = {"world": world, "fields_per_span": fields, "begin_ms": round(med(t_begin), 4), "exchange_ms": round(med(t_xchg), 4),
       "finish_ms": round(med(t_finish), 4), "span_airtime_ms": round(fields * 24.197, 1)}
if rank == 0:
    print(json.dumps(out))
if world > 1 or "RANK" in os.environ:
    dist.destroy_process_group()
