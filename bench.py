#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X 8VSB modulator hot path (BASELINE.json metric).

One "step" = one pass of the whole chain (randomizer -> RS -> interleaver -> trellis/map/sync ->
mixer+FIR) over one batch of synthetic TS.

--mode channels (default, BASELINE config 4): `--channels` independent channels x `--fields` fields per
  GPU (default 64 x 64 = the per-GPU shard of "512 channels over 8 GPUs").  Channels shard across ranks
  with no data-path collective (weak scaling): value = symbols of all ranks / max-over-ranks time.
--mode timesplit (BASELINE config 5): ONE transport stream (or `--channels` of them) cut into contiguous spans of
  `--fields` fields, one span per rank and step; the ranks hand the trellis state over with one 16-byte-per-channel
  all-gather (RCCL over xGMI), interleaver history comes from a one-field TS halo.  Before the timed region the
  spans of the first step are gathered on rank 0 and compared bit for bit (symbols and IQ) with a single-handle
  encode of the same stream.

Inputs are resident in HBM before the timed region and IQ is written to a resident HBM buffer (device
sink); nothing crosses PCIe inside the timed region.

Launching: `python bench.py --gpus N` starts N rank processes itself (one per GPU, fresh children, RCCL
rendezvous on 127.0.0.1); under torchrun (WORLD_SIZE set) it is one of the ranks.  It refuses (non-zero exit)
to run N ranks on fewer than N devices.  `--cpu-stub` exercises the same launcher / rendezvous / barrier /
max-over-ranks plumbing over gloo with the GPU step stubbed out (CPU tests only; its line says "stub": true).

Prints ONE JSON line (rank 0).  `roofline` is the mixer+FIR kernel timed with HIP events recorded
on the launch stream inside the timed region; `cpu_baseline` times the compiled reference
(oracle/_ref, kind "reference") or, if that library is absent, the C restatement (kind "port") on
the host cores of this box, rank 0 at N=1 only, on a bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SYMBOL_RATE_MSPS = 4500000.0 / 286 * 684 / 1e6  # 10.762238 Msym/s = one real-time channel
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak
CHAIN_BYTES_PER_SYM = 8.225    # whole chain fused minimum (58656 B in + 2083328 B out per field)
FIELD_SYMBOLS = 313 * 832
# (the line the driver keeps is its stdout TAIL: every string below is kept short so that the whole line stays under 4 KB)
FUSED_DTYPE = ("f32 (fp32 oscillator and accumulation; products on two fp16 pieces per operand, matrix cores, low-order pieces "
               "left out of the outermost taps: <=1.7e-6 measured, 7.2e-6 worst case with every term at its bound, tolerance 1e-5)")
WIDE_DTYPE = ("f32 (fp32 oscillator and accumulation; tap x oscillator operand in two fp16 pieces per tile, levels exact in one, "
              "matrix cores: 1.2e-6 measured, 8.4e-6 worst case with every term at its bound, tolerance 1e-5)")
METRIC = "real-time ATSC channels per GPU (Msym/s; 10.76=1 ch) at 1/2/4/8 MI355X"
PROFILE_ROUND = "r03"
PROFILE_DIR = os.path.join(ROOT, "profiles", PROFILE_ROUND)
KERNEL_SOURCES = ("kernels.hip", "fir_fft.hip", "fir_fused.hip", "fir_w.hip", "resample.hip", "static_entry.hip", "kernels.h",
                  "trellis_bits.h", "iq_format.h", "front_roles.h")


def kernel_sources_hash():
    """sha256 over the device sources of the product library (what a committed profile was taken from)."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        path = os.path.join(ROOT, "atsc_amd", "csrc", name)
        if os.path.exists(path):
            h.update(name.encode() + b"\0" + open(path, "rb").read())
    return h.hexdigest()


def kernel_sources_changed_since_profile():
    """True / False against profiles/rNN/kernel_sources.sha256 (written by scripts/summarize_profile.py when the counter
    passes are summarised); None when the round has no committed profile yet."""
    try:
        stored = open(os.path.join(PROFILE_DIR, "kernel_sources.sha256")).read().split()[0]
    except (OSError, IndexError):
        return None
    return stored != kernel_sources_hash()


def host_cores():
    """CPU cores this process may really use: the affinity mask, capped by a cgroup quota and by the box's CPU share
    (the pool exports it as OMP_NUM_THREADS; os.cpu_count() there shows the whole machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    why = "affinity mask"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max" and int(quota) // int(period) >= 1 and int(quota) // int(period) < n:
            n, why = int(quota) // int(period), "cgroup cpu.max"
    except (OSError, ValueError):
        pass
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and 0 < int(share) < n:
        n, why = int(share), "the box's CPU share (OMP_NUM_THREADS)"
    return n, why


def cpu_baseline(sample_fields=400):
    """Reference CPU encoder on this box's host cores (bounded sample: null-packet TS, null sink)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    use_ref = ol.ref_available()
    kind = "reference" if use_ref else "port"
    ncores, why = host_cores()
    ts = ol.ts_null(sample_fields)
    if not use_ref:
        sample_fields = max(4, sample_fields // 5)  # the scalar port is ~5x slower
        ts = ts[:sample_fields * ol.FIELD_TS]

    def work(out, i):
        enc = ol.Ref() if use_ref else ol.Oracle()
        t0 = time.perf_counter()
        if use_ref:
            # public API path, one field per call like util/atsc_encoder_null.cpp:40-43
            for f in range(sample_fields):
                enc.L.ref_process(enc.h, ts[f * ol.FIELD_TS:].ctypes.data_as(ol._u8p), 312, None, 0)
        else:
            enc.process_fields(ts, sample_fields, want=("iq",))
        out[i] = time.perf_counter() - t0

    one = [0.0]
    work(one, 0)
    single = sample_fields * ol.FIELD_SYMS / one[0] / 1e6
    outs = [0.0] * ncores
    th = [threading.Thread(target=work, args=(outs, i)) for i in range(ncores)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    multi = ncores * sample_fields * ol.FIELD_SYMS / wall / 1e6
    return {"value": round(multi, 2), "unit": "Msym/s", "cores": ncores, "kind": kind,
            "cores_from": why, "cpus_visible": os.cpu_count(),
            "single_thread_msym_s": round(single, 2),
            "channels_equiv": round(multi / SYMBOL_RATE_MSPS, 2),
            "sample": f"{sample_fields} fields of null packets per thread, null sink, {ncores} threads (one encoder each) + a 1-thread run"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", choices=("channels", "timesplit"), default="channels")
    ap.add_argument("--channels", type=int, default=None, help="channels per GPU (default 64; 1 in timesplit mode)")
    ap.add_argument("--fields", type=int, default=64, help="fields per channel per step (per rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--symbol-tap", action="store_true",
                    help="also write the u8 symbol tap (the parity check point) inside the timed region; the reference's "
                         "process() hands out IQ only (atsc.cpp:89), so the default workload does not")
    ap.add_argument("--iq-format", choices=("cf32", "cs16", "cs8"), default="cf32",
                    help="IQ sample format of the device sink; cf32 is the reference's payload and the headline "
                         "configuration, cs16/cs8 are the SDR sink formats (SURVEY 8f-3)")
    ap.add_argument("--fir-form", choices=("auto", "fft", "mfma", "fused", "wide"), default="auto",
                    help="A/B runs: which kernel computes the mixer + FIR (atsc_hip_set_fir_form)")
    ap.add_argument("--resample", default=None, metavar="L/M",
                    help="SURVEY 8f-1: follow the encoder with the SDR-rate polyphase resampler (include/atsc_hip_resample.h) "
                         "inside the timed step; the line gains a `resampler` object with that kernel's own roofline.  Not "
                         "the headline configuration (the reference runs the radio at the symbol rate)")
    ap.add_argument("--upsample", type=int, choices=(1, 2), default=1,
                    help="SURVEY 8f-1: 2 = the interpolating form of the filter (atsc_hip_set_upsample): two samples per symbol "
                         "out of the same pass, the even ones the reference's.  Not the headline configuration")
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="after the headline loop: this many seconds of back-to-back steps, reported as `sustained` (0 = skip)")
    ap.add_argument("--verbose-line", action="store_true", help="keep the replayed counter summaries in the line (long)")
    ap.add_argument("--settle-seconds", type=float, default=0.12,
                    help="untimed steps in front of the warm-up until the power manager has settled (0: none)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the legs behind the headline loop (sustained run, fp32-form run, in-run IQ error): profiler passes")
    ap.add_argument("--rccl", action="store_true",
                    help="run under this script's own launcher even with --gpus 1, so that the rank initialises an RCCL "
                         "process group and the collectives really run (backend \"nccl\" in the line)")
    ap.add_argument("--cpu-stub", action="store_true",
                    help="CPU tests: launcher + gloo rendezvous + barrier/max-over-ranks plumbing with the GPU step stubbed")
    args = ap.parse_args(argv)
    if args.channels is None:
        args.channels = 1 if args.mode == "timesplit" else 64
    return args


# ---------------------------------------------------------------------------------------------- launcher
def launch_ranks(args, argv):
    """`--gpus N` without a launcher around us: start N fresh rank processes (this process never touches the GPU)."""
    n = args.gpus
    if not args.cpu_stub:
        import torch
        # (on this image torch counts devices through amdsmi when it is importable and falls back to hipGetDeviceCount,
        # which does initialise the runtime in THIS process; either way this process starts no GPU work and the ranks are
        # fresh children)
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write(f"bench.py: --gpus {n} but only {have} GPU(s) are visible: refusing to report a "
                             f"{n}-GPU number from fewer devices\n")
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ATSC_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0 inherits stdout (the JSON line); the other ranks' stdout goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    deadline = time.time() + 3000
    for p in procs:
        try:
            code = p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            code = -9
        if code != 0 and rc == 0:
            rc = code if code > 0 else 1
            for q in procs:          # one rank failed: the others would wait in a collective for ever
                if q.poll() is None:
                    q.kill()
    if rc:
        sys.stderr.write(f"bench.py: a rank failed (exit {rc}); no result line is valid\n")
    return rc


class Ranks:
    """RANK / WORLD_SIZE plumbing shared by the GPU path and the CPU stub."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit(f"WORLD_SIZE={self.world} != --gpus {args.gpus}")
        self.dist = None
        self.backend = None

    def init(self, backend, device=None):
        under_launcher = "TORCHELASTIC_RUN_ID" in os.environ or "ATSC_BENCH_SELF_LAUNCHED" in os.environ
        if self.world > 1 or under_launcher:   # under a launcher the collective path runs even with one rank
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            kw = {"device_id": device} if device is not None else {}
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self.dist, self.backend = dist, backend

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, x, device="cpu"):
        if self.dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device=device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def ranks_seen(self, device="cpu"):
        """How many ranks, and which device ordinals, the collective backend really connected."""
        if self.dist is None:
            return 1, [self.local_rank]
        import torch
        mine = torch.tensor([self.local_rank], dtype=torch.int32, device=device)
        got = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(got, mine)
        return self.dist.get_world_size(), [int(g.item()) for g in got]

    def finish(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


LINE_LIMIT = 3900   # the driver keeps the TAIL of stdout: a line longer than its window would lose its head (metric, value)


def emit(json_fd, out):
    """ONE line.  If it outgrows LINE_LIMIT the explanatory keys go first (notes, then the replayed-profile pointers, then the
    secondary stage objects); the contract keys and the measured numbers stay."""
    def drop_notes(o):
        if isinstance(o, dict):
            return {k: drop_notes(v) for k, v in o.items() if k not in ("note", "sample", "measured", "handoff_note")}
        return o
    line = json.dumps(out)
    for shrink in (drop_notes,
                   lambda o: {k: v for k, v in o.items() if k != "roofline_other_stages"},
                   lambda o: {k: v for k, v in o.items() if k not in ("fp32_form", "rank_devices")}):
        if len(line) <= LINE_LIMIT:
            break
        out = shrink(out)
        out["line_shortened"] = True
        line = json.dumps(out)
    sys.stdout.flush()
    os.write(json_fd, (line + "\n").encode())


def run_stub(args, json_fd):
    """The rank plumbing with the GPU step replaced by a 1 ms sleep (gloo, CPU)."""
    rk = Ranks(args)
    rk.init("gloo")
    for _ in range(args.warmup):
        time.sleep(0.001)
    rk.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    rk.barrier()
    elapsed = rk.max_over_ranks(time.perf_counter() - t0)
    seen, ordinals = rk.ranks_seen()
    sym = args.channels * args.fields * FIELD_SYMBOLS * args.steps * rk.world
    if rk.rank == 0:
        emit(json_fd, {"metric": METRIC, "value": round(sym / elapsed / 1e6, 2), "unit": "Msym/s", "n_gpus": rk.world,
                       "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                       "stub": True, "ranks_seen": seen, "rank_devices": ordinals, "backend": "gloo",
                       "config": {"workload": "CPU stub of the rank plumbing (no GPU work)"}})
    rk.finish()


# ---------------------------------------------------------------------------------------------- GPU ranks
def load_profile_json(name):
    try:
        return json.load(open(os.path.join(PROFILE_DIR, name)))
    except (OSError, ValueError):
        return None


def make_ts(torch, channels, fields, seed):
    """Synthetic 19.39 Mbps TS per channel: PRBS payload behind a 0x47 sync byte, every 5th packet null-like."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    ts = torch.randint(0, 256, (channels, fields * 312, 188), dtype=torch.uint8, device="cuda", generator=g)
    ts[:, :, 0] = 0x47
    ts[:, ::5, 1:] = 0xFF          # the 'mix' flavour of config 2
    return ts.contiguous()


def run_gpu(args, json_fd):
    import torch
    import atsc_amd

    rk = Ranks(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the 8VSB path has no CPU fallback")
    if rk.local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rk.rank}: no GPU {rk.local_rank} on this node ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(rk.local_rank)
    rk.init("nccl", torch.device("cuda", rk.local_rank))
    seen, ordinals = rk.ranks_seen("cuda")
    if seen != args.gpus or len(set(ordinals)) != args.gpus:
        raise SystemExit(f"RCCL connected {seen} rank(s) on devices {ordinals}, wanted {args.gpus} distinct GPUs")

    fmt = {"cf32": atsc_amd.IQ_CF32, "cs16": atsc_amd.IQ_CS16, "cs8": atsc_amd.IQ_CS8}[args.iq_format]
    form = {"auto": atsc_amd.FIR_AUTO, "fft": atsc_amd.FIR_FFT, "mfma": atsc_amd.FIR_MFMA,
            "fused": atsc_amd.FIR_FUSED, "wide": atsc_amd.FIR_WIDE}[args.fir_form]
    sample_bytes = {"cf32": 8, "cs16": 4, "cs8": 2}[args.iq_format]
    if args.mode == "timesplit":
        res = bench_timesplit(args, rk, torch, atsc_amd, fmt, form, sample_bytes)
    else:
        res = bench_channels(args, rk, torch, atsc_amd, fmt, form, sample_bytes)
    res.update({"ranks_seen": seen, "rank_devices": ordinals, "backend": rk.backend or "none"})
    if rk.rank == 0:
        if rk.world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the checker must never take the bench down
                res["cpu_baseline"] = {"value": None, "error": repr(e)}
        emit(json_fd, res)
    rk.finish()


def stage_report(args, atsc_amd, enc, C_, F, fused, sample_bytes, msym_s, world, up=1, wide=False, other_stages=None):
    """Per-stage kernel time and the roofline objects.  The mixer + FIR kernel's time comes from the HIP events of the TIMED
    region (this rank's GPU); the timed region records those two events only (an event between two kernels costs the stream
    about 5 us: atsc_hip_timing_select), the other stages' times come from `other_stages`, a short pass with every stage
    timed that runs right behind it."""
    stage_ms = {k: 0.0 for k in atsc_amd.STAGE_NAMES}
    for b in range(args.steps):
        tm = enc.batch_timings(b)
        for k in stage_ms:
            stage_ms[k] += tm[k] / args.steps
    if other_stages:
        for k, v in other_stages.items():
            if k != "mix_fir":
                stage_ms[k] = v
    sym_per_step_gpu = C_ * F * FIELD_SYMBOLS
    # fused: 64584 interleaved bytes per field in (0.248 B/symbol) + IQ out; separate kernels: 1 B symbol in + IQ out
    fir_bytes_per_sym = (64584.0 / FIELD_SYMBOLS if fused else 1.0) + sample_bytes * up
    if fused and args.symbol_tap:
        fir_bytes_per_sym += 1.0   # the fused kernel also writes the u8 symbols
    fir_s = stage_ms["mix_fir"] / 1e3
    fir_gbs = fir_bytes_per_sym * sym_per_step_gpu / fir_s / 1e9 if fir_s > 0 else 0.0

    # HBM-bound front stages against the HBM roof with SURVEY 8(d)'s algorithmic bytes per field; the RS stage is bound
    # by LDS table lookups: its utilisation figures come from the committed counter passes
    alg_bytes_field = {"interleave": 2 * 64584, "trellis_map": 64584 + 260416}
    stages = {}
    for k, bpf in alg_bytes_field.items():
        if stage_ms[k] > 0:
            gbs = bpf * C_ * F / (stage_ms[k] / 1e3) / 1e9
            stages[k] = {"bound": "hbm", "algorithmic_bytes_per_launch": bpf * C_ * F, "avg_launch_ms": round(stage_ms[k], 4),
                         "achieved_gbs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
    stages["rand_rs"] = {"bound": "lds", "avg_launch_ms": round(stage_ms["rand_rs"], 4),
                         "gf256_mul_adds_per_launch": 3740 * 312 * C_ * F}
    # figures replayed from the committed counter passes (the driver's run cannot run rocprofv3): kept under their own
    # key, with the commit they were taken on and whether any kernel source has changed since
    replayed = {"dir": f"profiles/{PROFILE_ROUND}", "kernel_sources_changed_since_profile": kernel_sources_changed_since_profile()}
    util = load_profile_json("utilisation.json")
    if util and util.get("workload") == f"{C_} channels x {F} fields":
        replayed["commit"] = util.get("commit")
        if args.verbose_line:   # the counter summaries themselves stay in the committed files (utilisation.json, stall_counters.md)
            replayed["utilisation"] = util.get("kernels", {})

    traffic, traffic_src = None, None
    tj = load_profile_json("traffic.json")
    if tj and tj.get("workload") == f"{C_} channels x {F} fields" and sample_bytes == 8 and up == 1 and \
            ("fused" in tj.get("kernel", "")) == fused:
        traffic = tj["traffic_bytes_per_launch"]
        traffic_src = {"file": f"profiles/{PROFILE_ROUND}/traffic.json", "commit": tj.get("commit"),
                       "kernel_sources_changed_since_profile": replayed["kernel_sources_changed_since_profile"],
                       "note": "replayed from the committed FETCH_SIZE / WRITE_SIZE passes, not re-measured in this run"}
    stages["_from_committed_profile"] = replayed
    roofline = {
        "kernel": ("k_mix_fir_w (trellis / map / sync + mixer + RRC FIR, oscillator folded into the tap operand)" if wide else
                   "k_mix_fir_mm<fused> (trellis / map / sync + mixer + RRC FIR as a banded GEMM on the matrix cores)" if fused else
                   "k_mix_fir_fft (mixer + RRC FIR, overlap-save FFT form)"),
        "bound": "hbm",
        "achieved": round(fir_gbs, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(fir_gbs / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "traffic_source": traffic_src,
        "algorithmic_bytes_per_launch": fir_bytes_per_sym * sym_per_step_gpu,
        "avg_launch_ms": round(stage_ms["mix_fir"], 4),
        "note": f"{fir_bytes_per_sym:.3f} B/symbol algorithmic ("
                + ("0.248 B interleaved bytes in" if fused else "1 B symbol in")
                + f" + {up} x {sample_bytes} B {args.iq_format} out, SURVEY 8d); peak = 8 TB/s spec",
        "whole_chain_gbs": round((CHAIN_BYTES_PER_SYM - 8 + sample_bytes * up) * msym_s * 1e6 / world / 1e9, 1),
    }
    return stage_ms, roofline, stages


def iq_error_in_run(ts, first_iq, F):
    """Max |IQ - checker| of this run's own first batch (zero history), for the kept channels: the compiled reference
    when oracle/_ref travelled with the tree, else the C restatement.  The checker runs after the timed region."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol
    use_ref = ol.ref_available()
    worst, per = 0.0, {}
    t0 = time.perf_counter()
    for c, got in first_iq.items():
        h_ts = ts[c].reshape(-1).cpu().numpy()
        want = (ol.Ref() if use_ref else ol.Oracle()).process_fields(h_ts, F, want=("iq",))["iq"]
        e = float(np.abs(got - np.asarray(want, dtype=np.float32).reshape(-1)).max())
        per[str(c)] = e
        worst = max(worst, e)
    return {"max_abs_error": worst, "channels": sorted(int(c) for c in per), "fields": F, "tolerance": 1e-5,
            "against": "oracle/_ref (compiled reference)" if use_ref else "oracle/ (C restatement)",
            "measured": "in this run, first batch", "checker_seconds": round(time.perf_counter() - t0, 2)}


def bench_channels(args, rk, torch, atsc_amd, fmt, form, sample_bytes):
    C_, F = args.channels, args.fields
    enc = atsc_amd.Encoder(channels=C_, max_fields=F, device=rk.local_rank, timing=True)
    enc.set_iq_format(fmt)
    enc.set_fir_form(form)
    up = args.upsample
    if up != 1:
        if args.resample:
            raise SystemExit("--upsample and --resample are alternatives")
        enc.set_upsample(up)
    wide = enc.fir_form() == atsc_amd.FIR_WIDE
    fused = enc.fir_form() == atsc_amd.FIR_FUSED or wide
    ts = make_ts(torch, C_, F, 1234 + rk.rank)     # seeded per rank: every global channel its own stream
    sym = torch.empty(C_ * F * FIELD_SYMBOLS, dtype=torch.uint8, device="cuda") if args.symbol_tap else None
    stream = torch.cuda.current_stream().cuda_stream
    rs = None
    if args.resample:
        # the resampler reads complex64: the encoder hands it CF32 and the chosen sample format applies to ITS output
        L, M = (int(v) for v in args.resample.split("/"))
        enc.set_iq_format(atsc_amd.IQ_CF32)
        rs = atsc_amd.Resampler(C_, L, M, 16, out_format=fmt, device=rk.local_rank)
        n_in = F * FIELD_SYMBOLS
        n_out = max(rs.out_count(n_in), (n_in * L + M - 1) // M) + 8      # the count moves by one with the carried phase
        rs_out = torch.empty(C_ * n_out * sample_bytes, dtype=torch.uint8, device="cuda")
        rs_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        rs_samples = []
        iq_bytes = 8
    else:
        iq_bytes = sample_bytes
    iq = torch.empty(C_ * F * FIELD_SYMBOLS * iq_bytes * up, dtype=torch.uint8, device="cuda")
    rec = {"on": False, "k": 0}     # the resampler's events are recorded in the timed region only

    def step():
        enc.encode_batch(ts.data_ptr(), F, sym.data_ptr() if args.symbol_tap else None, iq.data_ptr(), stream)
        if rs is not None:
            k = rec["k"]
            on = rec["on"] and k < args.steps
            if on:
                rs_ev[k][0].record()
                rs_samples.append(rs.out_count(n_in))
            rs.resample(iq.data_ptr(), n_in, n_in, rs_out.data_ptr(), n_out, stream)
            if on:
                rs_ev[k][1].record()
                rec["k"] = k + 1

    # the handle's very first batch (zero history, field index 0): two channels of its IQ are kept on the host and
    # compared with the CPU checker after the timed region (`iq_max_abs_error_vs_oracle`)
    check_channels = sorted({0, C_ - 1})
    first_iq = None
    extras = rk.rank == 0 and not args.no_extras and rs is None
    if extras and iq_bytes == 8 and up == 1:
        step()
        torch.cuda.synchronize()
        per_ch = F * FIELD_SYMBOLS * 8
        first_iq = {c: iq[c * per_ch:(c + 1) * per_ch].cpu().numpy().view("float32") for c in check_channels}
    # Settle: the first ~15 steps after the GPU leaves idle run 5-25 % slow (the power manager overshoots, throttles, then
    # settles: per-step FIR times 1.97, 2.42, 2.30, 2.16, 2.07 ... 1.87 ms from a cold start); a modulator is a continuous load,
    # so the W warm-up steps and the K timed steps start from the settled state whatever W is.  Untimed, reported.
    settle_steps = 0
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < args.settle_seconds:
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        settle_steps += 8
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    enc.timing_reserve(args.steps)
    enc.timing_select(("mix_fir",))      # the timed region times the dominant kernel only
    rk.barrier()
    torch.cuda.synchronize()
    rec["on"] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    rk.barrier()
    elapsed = rk.max_over_ranks(time.perf_counter() - t0, "cuda")
    rec["on"] = False
    fir_timed = [enc.batch_timings(b) for b in range(args.steps)]
    if os.environ.get("ATSC_BENCH_DEBUG"):
        print("fir ms per timed step:", [round(t["mix_fir"], 4) for t in fir_timed], file=sys.stderr)
    # every stage timed, right behind the timed region (not part of `value`)
    n_all = max(5, min(args.steps, 20))
    enc.timing_select(None)
    enc.timing_reserve(n_all)
    for _ in range(n_all):
        step()
    torch.cuda.synchronize()
    other = {k: 0.0 for k in atsc_amd.STAGE_NAMES}
    for b in range(n_all):
        tm = enc.batch_timings(b)
        for k in other:
            other[k] += tm[k] / n_all
    enc.timing_select(("mix_fir",))

    sym_per_step_gpu = C_ * F * FIELD_SYMBOLS
    msym_s = sym_per_step_gpu * args.steps * rk.world / elapsed / 1e6

    class _Timed:   # stage_report reads the timed region's own records
        @staticmethod
        def batch_timings(b):
            return fir_timed[b]
    stage_ms, roofline, stages = stage_report(args, atsc_amd, _Timed, C_, F, fused, iq_bytes, msym_s, rk.world, up, wide, other)

    # ---- legs behind the headline loop (rank 0; the other ranks idle at the final barrier of run_gpu)
    sustained = fp32_form = None
    if extras:
        if args.sustain_seconds > 0:
            n = max(args.steps, int(args.sustain_seconds / (elapsed / args.steps)) + 1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            sustained = {"steps": n, "seconds": round(dt, 3), "ms_per_step": round(dt / n * 1e3, 4),
                         "value": round(sym_per_step_gpu * n / dt / 1e6, 2), "unit": "Msym/s (this GPU)",
                         "note": "back-to-back steps right behind the timed loop"}
        if fused and up == 1:
            # the same shard at the reference's arithmetic: fp32 throughout (k_trellis_map + k_mix_fir_fft)
            e32 = atsc_amd.Encoder(channels=C_, max_fields=F, device=rk.local_rank, timing=True)
            e32.set_iq_format(fmt)
            e32.set_fir_form(atsc_amd.FIR_FFT)
            n32 = max(5, min(args.steps, 20))
            for _ in range(3):
                e32.encode_batch(ts.data_ptr(), F, None, iq.data_ptr(), stream)
            torch.cuda.synchronize()
            e32.timing_reserve(n32)
            t1 = time.perf_counter()
            for _ in range(n32):
                e32.encode_batch(ts.data_ptr(), F, None, iq.data_ptr(), stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            st32 = {k: 0.0 for k in atsc_amd.STAGE_NAMES}
            for b in range(n32):
                tm = e32.batch_timings(b)
                for k in st32:
                    st32[k] += tm[k] / n32
            fir32 = st32["mix_fir"] / 1e3
            gbs32 = (1.0 + iq_bytes) * sym_per_step_gpu / fir32 / 1e9
            fp32_form = {"fir_form": "fft + k_trellis_map (fp32 throughout)",
                         "steps": n32, "ms_per_step": round(dt / n32 * 1e3, 4),
                         "value": round(sym_per_step_gpu * n32 / dt / 1e6, 2), "unit": "Msym/s (this GPU)",
                         "stage_ms": {k: round(st32[k], 4) for k in ("trellis_map", "mix_fir")},
                         "roofline": {"kernel": "k_mix_fir_fft", "bound": "hbm", "achieved": round(gbs32, 1), "peak": HBM_PEAK_GBS,
                                      "unit": "GB/s", "frac": round(gbs32 / HBM_PEAK_GBS, 4),
                                      "note": f"1 B symbol in + {iq_bytes} B out per symbol"}}
            del e32
    if first_iq is not None:
        try:
            roofline["iq_max_abs_error_vs_oracle"] = iq_error_in_run(ts, first_iq, F)
        except Exception as e:  # the checker must never take the bench down
            roofline["iq_max_abs_error_vs_oracle"] = {"error": repr(e)}
    resampler = None
    if rs is not None:
        ms = sum(a.elapsed_time(b) for a, b in rs_ev) / args.steps
        out_per_ch = sum(rs_samples) / args.steps
        byts = C_ * (n_in * 8 + out_per_ch * sample_bytes)
        resampler = {"ratio": args.resample, "taps_per_phase": 16, "out_format": args.iq_format, "avg_launch_ms": round(ms, 4),
                     "out_msamples_s": round(C_ * out_per_ch / ms / 1e3, 1),
                     "roofline": {"bound": "hbm", "achieved": round(byts / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(byts / ms / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": round(byts),
                                  "note": "8 B complex64 in per symbol + out_format bytes per output sample; kernel time from "
                                          "events on the launch stream around each call"},
                     "parity": "unpinned (no reference counterpart); every L-th sample of an integer-L output is the encoder's "
                               "own IQ bit for bit (tests/test_resample.py)"}
    res = {
        "metric": METRIC, "value": round(msym_s, 2), "unit": "Msym/s", "n_gpus": rk.world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": (WIDE_DTYPE if wide else FUSED_DTYPE) if fused else "f32",
        "data": "synthetic",
        "config": {"workload": f"{C_} channels x {F} fields per GPU per step (BASELINE config 4 per-GPU shard), "
                               f"TS resident in HBM, {'symbol tap + ' if args.symbol_tap else ''}"
                               f"{'complex64' if sample_bytes == 8 else args.iq_format} IQ to a device-resident sink",
                   "channels_per_gpu": C_, "fields_per_step": F,
                   "parallelism": f"channel-sharded x{rk.world}, no collective"},
        "realtime_channels": round(msym_s / SYMBOL_RATE_MSPS, 1),
        "realtime_channels_per_gpu": round(msym_s / SYMBOL_RATE_MSPS / rk.world, 1),
        "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
        "stage_ms_from": "mix_fir: HIP events of the timed region (the only events it records); others: every-stage pass behind it",
        "settle": {"untimed_steps_before_warmup": settle_steps, "why": "first ~15 steps after idle run 5-25 % slow (power transient)"},
        "roofline": roofline,
        "roofline_other_stages": stages,
    }
    if sustained:
        res["sustained"] = sustained
    if fp32_form:
        res["fp32_form"] = fp32_form
    res["kernel_sources_changed_since_profile"] = stages.get("_from_committed_profile", {}).get("kernel_sources_changed_since_profile")
    if up != 1:
        res["upsample"] = {"factor": up, "out_msamples_s": round(msym_s * up, 1),
                           "parity": "even samples: the reference's IQ bit for bit (pinned by the oracle); odd samples unpinned (no "
                                     "reference counterpart), within 1e-5 of a float64 model of the formula (tests/test_gpu_upsample.py)"}
        res["config"]["workload"] += f"; interpolating filter, {up} samples per symbol out of the same pass"
    if resampler:
        res["resampler"] = resampler
        res["config"]["workload"] += f" + {args.resample} SDR-rate resampler ({args.iq_format} out) inside the step"
    return res


def bench_timesplit(args, rk, torch, atsc_amd, fmt, form, sample_bytes):
    """BASELINE config 5: `channels` streams, each cut into spans of F fields; in step k rank r encodes the span that
    starts at field (k * world + r) * F.  A step = prepare (halo field + state import at the span's field index) +
    encode_begin + the parity all-gather + encode_finish."""
    from atsc_amd.timesplit import TimeSplitEncoder, exclusive_xor_over_ranks
    C_, F, W = args.channels, args.fields, rk.world
    n_span = F * FIELD_SYMBOLS
    # The stream: every rank generates the same fields [0, W*F) from one seed (only a one-field halo is ever needed
    # from the neighbour, and it is part of the stream: no transfer).  Steps re-use these fields at later field
    # indices (the TS content does not matter for timing; field parity and oscillator phase follow the index).
    # With ONE rank the check would be vacuous (a span that starts at field 0 needs no halo, no seek and no carry), so the
    # stream is then two spans long and this rank encodes both, the second on a handle of its own that starts mid-stream
    # with the first span's parity as its carry: halo, seek and hand-off are all compared.
    S = W if W > 1 else 2
    ts_all = make_ts(torch, C_, S * F, 4242).view(C_, S * F, 312 * 188)
    my_ts = ts_all[:, rk.rank * F:(rk.rank + 1) * F].contiguous()
    halo = ts_all[:, rk.rank * F - 1].contiguous() if rk.rank else None
    any_halo = halo if halo is not None else ts_all[:, 0].contiguous()   # later steps: every rank starts mid-stream
    d_sym = torch.empty(C_ * n_span, dtype=torch.uint8, device="cuda")
    d_iq = torch.empty(C_ * n_span * sample_bytes, dtype=torch.uint8, device="cuda")
    group = None

    # ---- parity of the first step against ONE handle encoding the whole stream (rank 0 holds the comparison)
    tse = TimeSplitEncoder(F, device=rk.local_rank, channels=C_, fir_form=form, iq_format=fmt)
    tse.prepare(rk.rank * F, halo)
    par0 = tse.encode(my_ts, d_sym, d_iq, group)
    torch.cuda.synchronize()
    parity = {"checked": False}
    check_fields = S * F
    if check_fields * C_ * FIELD_SYMBOLS * (1 + sample_bytes) <= 64e9:
        if W > 1:
            syms = [torch.empty_like(d_sym) for _ in range(W)] if rk.rank == 0 else None
            iqs = [torch.empty_like(d_iq) for _ in range(W)] if rk.rank == 0 else None
            rk.dist.gather(d_sym, syms, dst=0)
            rk.dist.gather(d_iq, iqs, dst=0)
        else:
            tse1 = TimeSplitEncoder(F, device=rk.local_rank, channels=C_, fir_form=form, iq_format=fmt)
            tse1.prepare(F, ts_all[:, F - 1].contiguous())
            sym1, iq1 = torch.empty_like(d_sym), torch.empty_like(d_iq)
            tse1.encode(ts_all[:, F:2 * F].contiguous(), sym1, iq1, group, exchange=lambda par, grp: par0.clone())
            torch.cuda.synchronize()
            syms, iqs = [d_sym, sym1], [d_iq, iq1]
            del tse1
        if rk.rank == 0:
            whole = atsc_amd.Encoder(channels=C_, max_fields=S * F, device=rk.local_rank)
            whole.set_iq_format(fmt)
            whole.set_fir_form(tse.enc.fir_form())
            w_sym = torch.empty(C_ * S * n_span, dtype=torch.uint8, device="cuda")
            w_iq = torch.empty(C_ * S * n_span * sample_bytes, dtype=torch.uint8, device="cuda")
            whole.encode_batch(ts_all.data_ptr(), S * F, w_sym.data_ptr(), w_iq.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            w_sym = w_sym.view(C_, S, n_span)
            w_iq = w_iq.view(C_, S, n_span * sample_bytes)
            sym_bad = sum(int((syms[r].view(C_, n_span) != w_sym[:, r]).sum()) for r in range(S))
            iq_bad = sum(int((iqs[r].view(C_, n_span * sample_bytes) != w_iq[:, r]).sum()) for r in range(S))
            parity = {"checked": True, "fields": check_fields, "spans": S, "symbol_mismatches": sym_bad, "iq_byte_mismatches": iq_bad,
                      "against": "one handle encoding the whole stream on rank 0 (atsc_hip_encode_batch)"
                                 + ("" if W > 1 else "; one rank: both spans encoded here, the second from a one-field halo, a seek "
                                                     "to its first field and the first span's parity as carry")}
            del whole, w_sym, w_iq, syms, iqs
            if sym_bad or iq_bad:
                raise SystemExit(f"time-split output differs from the continuous encode: {parity}")
    rk.barrier()

    # ---- timed region
    t_exchange = []

    def timed_exchange(par, grp):
        torch.cuda.synchronize()            # the parity is ready: what follows is the hand-off itself
        t0 = time.perf_counter()
        carry = exclusive_xor_over_ranks(par, grp)
        torch.cuda.synchronize()
        t_exchange.append(time.perf_counter() - t0)
        return carry

    def step(k, exchange):
        first = (k * W + rk.rank) * F
        tse.prepare(first, any_halo if first else None)
        tse.encode(my_ts, None if not args.symbol_tap else d_sym, d_iq, group, exchange=exchange)

    for k in range(args.warmup):
        step(1 + k, exclusive_xor_over_ranks)
    torch.cuda.synchronize()
    rk.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(1 + args.warmup + k, timed_exchange)
    torch.cuda.synchronize()
    rk.barrier()
    elapsed = rk.max_over_ranks(time.perf_counter() - t0, "cuda")
    handoff_us = rk.max_over_ranks(sorted(t_exchange)[len(t_exchange) // 2] * 1e6 if t_exchange else 0.0, "cuda")

    msym_s = C_ * n_span * args.steps * W / elapsed / 1e6
    return {
        "metric": METRIC, "value": round(msym_s, 2), "unit": "Msym/s", "n_gpus": W, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if tse.enc.fir_form() == atsc_amd.FIR_FFT else FUSED_DTYPE, "data": "synthetic",
        "config": {"workload": f"BASELINE config 5: {C_} transport stream(s) time-split at field boundaries, {W} span(s) of "
                               f"{F} fields per step (one per GPU), trellis state handed over by a {16 * C_}-byte all-gather, "
                               f"interleaver history from a one-field TS halo; {args.iq_format} IQ to a device-resident sink",
                   "channels": C_, "fields_per_span": F, "parallelism": f"time-split x{W}, one all-gather per step"},
        "realtime_channels": round(msym_s / SYMBOL_RATE_MSPS, 1),
        "handoff_latency_us_median": round(handoff_us, 1),
        "handoff_note": ("host-timed: device sync, all-gather of the span parities over RCCL (device tensors), XOR fold, device "
                         "sync; max over ranks of each rank's median" if rk.backend == "nccl" else
                         "no collective ran (single process, no process group): the figure is two device syncs around a "
                         "zero-fill, not a hand-off; run with --rccl (or under torchrun) for an RCCL-backed figure"),
        "parity": parity,
        "fir_form": {atsc_amd.FIR_FFT: "fft", atsc_amd.FIR_MFMA: "mfma", atsc_amd.FIR_FUSED: "fused",
                     atsc_amd.FIR_WIDE: "wide"}[tse.enc.fir_form()],
    }


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if (args.gpus > 1 or args.rccl) and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))

    # stdout carries exactly one JSON line: whatever libraries print there (RCCL's version banner, ...) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.cpu_stub:
        run_stub(args, json_fd)
    else:
        run_gpu(args, json_fd)


if __name__ == "__main__":
    main()
