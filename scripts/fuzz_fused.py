#!/usr/bin/env python3
"""Randomised cross-check of the fused kernel against the separate kernels (run on the GPU box):
random channel counts, field counts and batch sequences; per case the fused kernel's symbol tap must equal
k_trellis_map's symbols and its complex64 IQ must equal the unfused matrix-core form's bit for bit, over consecutive
batches of one handle (carried state); CS16 / CS8 must equal the documented formula applied to that IQ.
  python scripts/fuzz_fused.py [cases=12] [seed=1]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import atsc_amd  # noqa: E402
from atsc_amd import FIELD_SYMBOLS  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda")
for case in range(cases):
    C_ = int(rng.integers(16, 81))
    nf = int(rng.integers(1, 6))
    batches = int(rng.integers(1, 4))
    g.manual_seed(1000 + case)
    encs = {}
    for name, form, fmt in (("mfma", atsc_amd.FIR_MFMA, atsc_amd.IQ_CF32), ("fused", atsc_amd.FIR_FUSED, atsc_amd.IQ_CF32),
                            ("cs16", atsc_amd.FIR_FUSED, atsc_amd.IQ_CS16), ("cs8", atsc_amd.FIR_FUSED, atsc_amd.IQ_CS8)):
        e = atsc_amd.Encoder(channels=C_, max_fields=5)
        e.set_fir_form(form)
        e.set_iq_format(fmt)
        encs[name] = e
    for b in range(batches):
        f = int(rng.integers(1, nf + 1))           # batches of different lengths on one handle
        n = C_ * f * FIELD_SYMBOLS
        ts = torch.randint(0, 256, (C_, f * 312, 188), dtype=torch.uint8, device="cuda", generator=g)
        ts[:, :, 0] = 0x47
        if rng.integers(0, 2):
            ts[:, ::3, 1:] = 0xFF
        out = {}
        for name, e in encs.items():
            sb = {"mfma": 8, "fused": 8, "cs16": 4, "cs8": 2}[name]
            d_sym = torch.full((n + 256,), 0xEE, dtype=torch.uint8, device="cuda")
            d_iq = torch.full((n * sb + 256,), 0x5A, dtype=torch.uint8, device="cuda")
            tap = name in ("mfma", "fused") or bool(rng.integers(0, 2))
            e.encode_batch(ts.data_ptr(), f, d_sym.data_ptr() if tap else None, d_iq.data_ptr(), st)
            torch.cuda.synchronize()
            assert bool((d_iq[n * sb:] == 0x5A).all()) and bool((d_sym[n:] == 0xEE).all()), (case, b, name, "wrote past the end")
            out[name] = (d_sym[:n] if tap else None, d_iq[:n * sb])
        assert bool((out["mfma"][0] == out["fused"][0]).all()), (case, b, "symbols")
        assert bool((out["mfma"][1] == out["fused"][1]).all()), (case, b, "IQ bits")
        y = out["fused"][1].view(torch.float32).double().clamp(-1.0, 1.0)
        for name, full, dt in (("cs16", 32767.0, torch.int16), ("cs8", 127.0, torch.int8)):
            want = torch.round(y * full)            # ties to even, like the kernel
            got = out[name][1].view(dt).double()
            assert bool((got == want).all()), (case, b, name, int((got != want).sum()))
            if out[name][0] is not None:
                assert bool((out[name][0] == out["mfma"][0]).all()), (case, b, name, "symbols")
    print(f"case {case}: {C_} channels, batches of up to {nf} fields x {batches}: ok", flush=True)
print("all ok")
