#!/usr/bin/env python3
"""Summarise a gpurun_out/prof_<tag> directory (made by scripts/profile_gpu.sh) into profiles/<name>/:
  kernel_stats.csv   rocprofv3 --kernel-trace --stats of the default bench.py run, as written by the profiler
  counters.csv       every collected counter, mean per launch, per kernel of this library
  traffic.json       HBM bytes per launch of the mixer + FIR kernel (what bench.py reports as roofline.traffic)
  utilisation.json   per kernel: LDS-array / VALU / MFMA busy fractions and the LDS bank-conflict share
usage: scripts/summarize_profile.py gpurun_out/prof_r02 profiles/r02 [commit]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else None
WORKLOAD = "64 channels x 64 fields"
CUS, XCDS = 256, 8
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, "kernel_stats.csv"))
if os.path.exists(os.path.join(src, "bench_under_trace.json")):
    shutil.copy(os.path.join(src, "bench_under_trace.json"), os.path.join(dst, "bench_under_trace.json"))
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
mean = {k: {n: sum(v) / len(v) for n, v in acc[k].items() if v} for k in acc}
names = sorted({c for k in acc for c in acc[k]})
ours = [k for k in sorted(acc) if k.startswith("atsc_k::")]
with open(os.path.join(dst, "counters.csv"), "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["kernel", "launches(trace run)", "avg_us(trace run)", "min_us"] + names)
    for k in ours:
        d = dur.get(k, [0])
        w.writerow([k, len(d), round(sum(d) / len(d), 1), round(min(d), 1)] + [round(mean[k][n]) if n in mean[k] else "" for n in names])

# HBM traffic per launch of the dominant kernel, corrected as MI355X_MICROARCH.md prescribes for gfx950:
# FETCH_SIZE (KiB) counts 128-B read requests at 64 B -> x2 for wide coalesced reads; WRITE_SIZE (KiB) is exact.
fir = next((k for k in ours if k.startswith("atsc_k::k_mix_fir_w")), None) or \
      next((k for k in ours if k.startswith("atsc_k::k_mix_fir_mm<0, true")), None) or \
      next((k for k in ours if k.startswith("atsc_k::k_mix_fir")), None)
if fir and "FETCH_SIZE" in mean[fir] and "WRITE_SIZE" in mean[fir]:
    fetch, write = mean[fir]["FETCH_SIZE"] * 1024 * 2, mean[fir]["WRITE_SIZE"] * 1024
    json.dump({"kernel": fir + (" (the fused form)" if "true" in fir else ""), "fetch_bytes_per_launch": round(fetch),
               "write_bytes_per_launch": round(write), "traffic_bytes_per_launch": round(fetch + write),
               "workload": WORKLOAD, "commit": commit,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, scripts/profile_gpu.sh), FETCH_SIZE x2 (gfx950)"},
              open(os.path.join(dst, "traffic.json"), "w"), indent=1)

# Utilisation per kernel.  SQ counters are sums over the chip's CUs; a kernel's CU-cycles = GRBM_GUI_ACTIVE / 8 XCDs x 256.
#   lds_array_busy      = SQ_LDS_IDX_ACTIVE / CU-cycles               (cycles the LDS array works, conflicts included)
#   lds_conflict_share  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE    (of those, extra cycles from bank conflicts)
#   valu_issue          = 4 x SQ_INSTS_VALU / (4 SIMDs x CU-cycles)   (a wave64 vector instruction holds its SIMD 4 cycles; MFMAs longer)
#   mfma_busy           = SQ_VALU_MFMA_BUSY_CYCLES / (4 x CU-cycles)
stage_of = {"k_rs_parity": "rand_rs", "k_interleave": "interleave", "k_trellis_scan": "trellis_scan", "k_trellis_map": "trellis_map",
            "k_osc_table": "osc_table", "k_scan_osc_state": "osc_table", "k_mix_fir": "mix_fir"}
util = {}
for k in ours:
    m = mean[k]
    if "GRBM_GUI_ACTIVE" not in m:
        continue
    cu_cycles = m["GRBM_GUI_ACTIVE"] / XCDS * CUS
    stage = next((s for p, s in stage_of.items() if p in k), None)
    if not stage or stage in util:
        continue
    u = {"kernel": k, "cu_cycles_per_launch": round(cu_cycles)}
    if "SQ_LDS_IDX_ACTIVE" in m:
        u["lds_array_busy"] = round(m["SQ_LDS_IDX_ACTIVE"] / cu_cycles, 3)
        if m["SQ_LDS_IDX_ACTIVE"] > 0 and "SQ_LDS_BANK_CONFLICT" in m:
            u["lds_conflict_share"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 3)
    if "SQ_INSTS_VALU" in m:
        u["valu_issue"] = round(m["SQ_INSTS_VALU"] / cu_cycles, 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        u["mfma_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * cu_cycles), 3)
    if "SQ_WAIT_INST_ANY" in m and "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0:
        u["wave_cycles_waiting"] = round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 3)
    util[stage] = u
json.dump({"workload": WORKLOAD, "commit": commit, "source": "scripts/profile_gpu.sh + scripts/summarize_profile.py (formulas there)",
           "kernels": util}, open(os.path.join(dst, "utilisation.json"), "w"), indent=1)

# where the FIR kernel's waves wait (store path)
if fir:
    rows = ["GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_DATA_FIFO_FULL",
            "SQ_LDS_CMD_FIFO_FULL", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "TA_TA_BUSY_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum",
            "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_PENDING_STALL_CYCLES_sum", "TD_TD_BUSY_sum",
            "TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_WRREQ_STALL_sum"]
    with open(os.path.join(dst, "stall_counters.md"), "w") as fo:
        fo.write(f"# {fir}: where the waves wait (scripts/profile_gpu.sh, rocprofv3 --pmc, mean per launch, {WORKLOAD}, commit {commit})\n\n")
        fo.write("Sums over all instances of a block on the chip (256 CUs / TA / TCP / TD, 128 TCC channels, 8 XCDs for GRBM).\n\n| Counter | value |\n|---|---|\n")
        for n in rows:
            if n in mean[fir]:
                fo.write(f"| {n} | {round(mean[fir][n]):,} |\n".replace(",", " "))
# what the passes were taken from: bench.py compares this with the tree it runs from (kernel_sources_changed_since_profile)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
open(os.path.join(dst, "kernel_sources.sha256"), "w").write(f"{bench.kernel_sources_hash()}  atsc_amd/csrc device sources ({', '.join(bench.KERNEL_SOURCES)}), commit {commit}\n")
print(open(os.path.join(dst, "counters.csv")).read())
print(open(os.path.join(dst, "utilisation.json")).read())
