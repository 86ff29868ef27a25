#!/usr/bin/env python3
"""Phase timestamps of one k_interleave workgroup (GPU box).  Prints mean cycles per phase per wave.

Needs a variant library build/var/il_ts (scripts/dev/variant.sh il_ts "" kernels.hip <copy>) from a COPY of kernels.hip in which
one workgroup (blockIdx (3, 5), chunks 8..40) writes __builtin_readcyclecounter() at the phase boundaries of chunk() into the
front of the `par` buffer, [chunk][wave][8 stamps] as uint64: the instrumented copy is not kept in the tree (it is the kernel
plus seven stores); DESIGN.md 3 (k_interleave) quotes what it showed."""
import ctypes, os, sys
import numpy as np
import torch
os.environ["ATSC_HIP_LIB"] = os.path.abspath("build/var/il_ts/libatsc_hip.so")
sys.path.insert(0, os.getcwd())
import atsc_amd
C_, F = 64, 64
enc = atsc_amd.Encoder(channels=C_, max_fields=F)
ts = torch.randint(0, 256, (C_, F * 312, 188), dtype=torch.uint8, device="cuda"); ts[:, :, 0] = 0x47
iq = torch.empty(C_ * F * atsc_amd.FIELD_SYMBOLS, dtype=torch.complex64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    enc.encode_batch(ts.data_ptr(), F, None, iq.data_ptr(), st)
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ["ATSC_HIP_LIB"])
buf = np.zeros(8 * 64 * 10, np.uint64)
assert L.atsc_dbg_il_ts(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(8, 64, 10).astype(np.int64)
names = ["fetch issue", "gather", "wait barrier B", "fold+stores", "parity", "commit (waits for the fetch)", "wait barrier C", "-"]
for w in range(8):
    d = np.diff(t[w, 8:40, :9], axis=1).mean(axis=0)
    per = (t[w, 9:40, 0] - t[w, 8:39, 0]).mean()
    print("wave", w, "chunk period", round(per), " ".join(f"{n}={round(x)}" for n, x in zip(names, d)))
