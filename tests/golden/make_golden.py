"""Generate the golden fixtures from the COMPILED REFERENCE (oracle/_ref/libatsc_ref.so).

Run in a container that has /root/reference (after `make -C oracle`):
    python tests/golden/make_golden.py
Outputs (committed): tests/golden/golden.json  (sha256 of full stage dumps, trellis carries, spot values)
                     tests/golden/windows.npz  (symbol and IQ windows for direct comparison)
Inputs are seeded (`null`, `mt1234`, see tests/oracle_lib.py), so they are not stored.
Fixtures are data only: inputs' recipe + expected outputs of the reference.
"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle_lib import FIELD_SYMS, Ref, ts_mix, ts_mt, ts_null  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def iq_window_index(nfields):
    """first/last 4096 samples of fields 0,1,3 plus every 97th sample of everything."""
    idx = []
    for f in (0, 1, 3):
        if f < nfields:
            idx.append(np.arange(f * FIELD_SYMS, f * FIELD_SYMS + 4096))
            idx.append(np.arange((f + 1) * FIELD_SYMS - 4096, (f + 1) * FIELD_SYMS))
    idx.append(np.arange(0, nfields * FIELD_SYMS, 97))
    return np.unique(np.concatenate(idx))


def main():
    meta = {"generator": "tests/golden/make_golden.py", "source": "compiled reference, FMA+BMI2 path, glibc libm",
            "cases": {}}
    win = {}
    for name, ts, nf in (("null4", ts_null(4), 4), ("mt1234_4", ts_mt(1234, 4), 4), ("mix7_3", ts_mix(7, 3), 3)):
        r = Ref()
        d = r.process_fields(ts, nf, want=("enc", "il", "sym", "iq"))
        iq = d["iq"].view(np.complex64)
        widx = iq_window_index(nf)
        meta["cases"][name] = {
            "fields": nf,
            "sha256": {k: sha(d[k]) for k in ("enc", "il", "sym", "iq")},
            "carry_z2_z0a_z0b": r.carry().tolist(),
            "first16_data_seg0": d["sym"][832:848].tolist(),
        }
        win[name + "_sym_head"] = d["sym"][:2 * 832].copy()          # field sync + first data segment, field 0
        win[name + "_sym_f1_head"] = d["sym"][FIELD_SYMS:FIELD_SYMS + 2 * 832].copy()  # odd field sync
        win[name + "_sym_tail"] = d["sym"][-832:].copy()
        win[name + "_sym_stride"] = d["sym"][::53].copy()
        win[name + "_iq_idx"] = widx.astype(np.int64)
        win[name + "_iq"] = iq[widx].copy()
    # public-API path, 64 fields in 1000-packet chunks
    ts = ts_mt(1234, 64)
    r = Ref()
    total, chunks = 0, []
    for off in range(0, 64 * 312, 1000):
        cnt = min(1000, 64 * 312 - off)
        n, iq = r.process(ts[off * 188:(off + cnt) * 188], cnt, 8)
        total += n
        chunks.append(iq)
    iq64 = np.concatenate(chunks)
    meta["cases"]["mt1234_64_api"] = {"fields": 64, "callbacks": total, "sha256": {"iq": sha(iq64)}}
    meta["osc"] = {
        "scale_bits": [int(x) for x in Ref.scale().view(np.uint32)],
        "T0_sha256": sha(Ref().osc_table()),
        "T0_spot_bits": {str(i): [int(x) for x in Ref().osc_table().view(np.uint32)[2 * i:2 * i + 2]]
                         for i in (0, 1, 1000, 260415, 260416, 261247)},
    }
    meta["taps_bits"] = [int(x) for x in Ref.taps().view(np.uint32)]
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "windows.npz"), **win)
    print(json.dumps({k: v["sha256"] for k, v in meta["cases"].items()}, indent=1))


if __name__ == "__main__":
    main()
