import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import atsc_amd
from atsc_amd import FIELD_SYMBOLS as FS
from oracle_lib import Oracle
C_, nf = 16, 1
rng = np.random.default_rng(1)
sym = rng.integers(0, 8, size=(C_, nf * FS), dtype=np.uint8)
enc = atsc_amd.Encoder(channels=C_, max_fields=nf)
enc.set_fir_form(atsc_amd.FIR_MFMA)
d_sym = torch.from_numpy(sym).cuda()
d_iq = torch.zeros(C_ * nf * FS, dtype=torch.complex64, device="cuda")
enc.mix_fir(d_sym.data_ptr(), nf, d_iq.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
iq = d_iq.cpu().numpy().view(np.float32).reshape(C_, nf * FS, 2)
for c in (0, 1, 15):
    want = Oracle().mix_fir(sym[c], nf).reshape(nf * FS, 2)
    err = np.abs(iq[c] - want)
    print("ch", c, "max err", err.max(), "first bad", np.argwhere(err > 1e-5)[:5].tolist())
    print(" got", iq[c, :4].tolist(), "\n want", want[:4].tolist())
    e1 = err.max(axis=1)
    print(" err by pos%32:", np.array([e1[k::32].max() for k in range(32)]).round(3).tolist())
    print(" err by tile (first 12):", np.array([e1[32*k:32*k+32].max() for k in range(12)]).round(3).tolist())
