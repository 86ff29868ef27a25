#!/usr/bin/env python3
"""Summarize a gpurun_out/prof_<tag> directory (made by scripts/profile_gpu.sh) into profiles/<name>/.
Copies the rocprofv3 kernel_stats CSV and writes one counters.csv (mean per launch, per kernel)."""
import collections
import csv
import glob
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, "kernel_stats.csv"))
trace = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))[0]
def kname(n):
    n = n.split("(")[0]
    return n[5:] if n.startswith("void ") else n      # template instances print their return type


dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    dur[kname(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
with open(os.path.join(dst, "counters.csv"), "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["kernel", "launches", "avg_us(trace run)", "min_us"] + names)
    for k in sorted(acc):
        if not k.startswith("atsc_k::"):
            continue
        d = dur.get(k, [0])
        w.writerow([k, len(d), round(sum(d) / len(d), 1), round(min(d), 1)] +
                   [round(sum(acc[k][n]) / len(acc[k][n])) if acc[k][n] else "" for n in names])
# HBM traffic per launch of the dominant kernel, corrected as MI355X_MICROARCH.md prescribes for gfx950:
# FETCH_SIZE (KiB) counts 128-B read requests at 64 B -> x2 for wide coalesced reads; WRITE_SIZE (KiB) is exact.
import json
k = next((n for n in acc if n.startswith(("atsc_k::k_mix_fir_fused<0>", "atsc_k::k_mix_fir_fft<0>"))), "atsc_k::k_mix_fir_fft")
if k in acc and acc[k].get("FETCH_SIZE") and acc[k].get("WRITE_SIZE"):
    fetch = sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"]) * 1024 * 2
    write = sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"]) * 1024
    json.dump({"kernel": k, "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
               "traffic_bytes_per_launch": round(fetch + write), "workload": "64 channels x 64 fields",
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950)"},
              open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, "counters.csv")).read())
