#!/usr/bin/env python3
"""Throughput of the SDR-rate resampler on modulator-shaped data: 64 channels x 8 fields of complex64,
2x (the common SDR case) and 8/7; HIP-event timed.  Prints one JSON line per case (not the headline metric)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import atsc_amd

C_, F = 64, 8
n = F * atsc_amd.FIELD_SYMBOLS
x = torch.randn(C_, n, 2, device="cuda", dtype=torch.float32) * 0.3
st = torch.cuda.current_stream().cuda_stream
for L, M, fmt, name in ((2, 1, atsc_amd.IQ_CF32, "cf32"), (2, 1, atsc_amd.IQ_CS16, "cs16"), (8, 7, atsc_amd.IQ_CF32, "cf32")):
    rs = atsc_amd.Resampler(C_, L, M, 16, out_format=fmt)
    esz = {"cf32": 8, "cs16": 4}[name]
    n_out = rs.out_count(n)
    y = torch.empty(C_ * (n_out + 8) * esz, dtype=torch.uint8, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(13):
        if it == 3:
            ev[0].record()
        assert rs.out_count(n) <= n_out + 8
        rs.resample(x.data_ptr(), n, n, y.data_ptr(), n_out + 8, st)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 10
    byts = C_ * (n * 8 + n_out * esz)
    print(json.dumps({"resample": f"{L}/{M}", "out": name, "ms": round(ms, 4), "in_msym_s": round(C_ * n / ms / 1e3, 1),
                      "algorithmic_gbs": round(byts / ms / 1e6, 1), "frac_of_hbm_peak": round(byts / ms / 1e6 / 8000, 3)}))
