# GPU box: what does a plain device copy of the interleaver's byte count reach?  (read N, write N)
import torch
n = 64 * 64 * 64584
src = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
dst = torch.empty_like(src)
for name, fn in (("tensor.copy_", lambda: dst.copy_(src)),):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"{name}: {n/1e6:.1f} MB in + out, {best*1e3:.1f} us, {2*n/best/1e9:.2f} TB/s")
