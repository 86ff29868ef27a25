#!/usr/bin/env python3
"""Does the front of one batch (RS + interleaver) run under the FIR of another?  Two handles, the same work either on one
stream (serial) or on two streams (free to overlap).  GPU box: python scripts/dev/overlap_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import atsc_amd
from atsc_amd import FIELD_SYMBOLS

C_, F, STEPS = 64, 64, 20
g = torch.Generator(device="cuda")
g.manual_seed(1)
ts = torch.randint(0, 256, (C_, F * 312, 188), dtype=torch.uint8, device="cuda", generator=g)
ts[:, :, 0] = 0x47
encs = [atsc_amd.Encoder(channels=C_, max_fields=F) for _ in range(2)]
iqs = [torch.empty(C_ * F * FIELD_SYMBOLS, dtype=torch.complex64, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(two_streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(STEPS):
        i = k & 1
        s = streams[i if two_streams else 0]
        encs[i].encode_batch(ts.data_ptr(), F, None, iqs[i].data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3


for _ in range(2):
    run(False), run(True)
print("ms per step: one stream %.4f, two streams %.4f" % (run(False), run(True)))
