"""CPU-side checks of the product boundary (no compute, no GPU needed):
the C-ABI library loads, exports every symbol include/atsc_hip.h declares, builds the constant tables
bit-identically to the oracle / golden fixtures, and refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import atsc_amd
from oracle_lib import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    """Every function declared in include/*.h (typedef'd callback types excluded)."""
    syms = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        txt = re.sub(r"typedef[^;{]*\([^;]*;", "", txt)
        syms |= set(re.findall(r"\b(atsc_(?:hip|ts)_[a-z_0-9]+)\s*\(", txt))
    return sorted(syms)


def test_library_exports_every_declared_symbol():
    lib = atsc_amd.load_library()
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/*.h but not exported"
    assert sorted(atsc_amd.EXPORTS) == syms, "python binding list and header disagree"


def test_header_cites_reference_for_each_entry_point():
    txt = open(os.path.join(ROOT, "include", "atsc_hip.h")).read()
    for anchor in ("src/atsc/atsc.h:7-16", "src/atsc/atsc.cpp:59-77", "src/atsc/atsc.cpp:93-95", "src/atsc/atsc.cpp:79-90"):
        assert anchor in txt


def test_struct_layout_matches_header():
    # atsc_hip_stream_state: u64 + 10608 + 12*4 bytes, 8-byte aligned
    assert C.sizeof(atsc_amd.StreamState) == (8 + 10608 + 48 + 7) // 8 * 8
    assert C.sizeof(atsc_amd.Config) == 16


def test_tables_match_oracle_and_golden(golden):
    g, _ = golden
    assert np.array_equal(atsc_amd.get_table(0).view(np.uint32), Oracle.taps().view(np.uint32))
    assert [int(x) for x in atsc_amd.get_table(0).view(np.uint32)] == g["taps_bits"]
    assert np.array_equal(atsc_amd.get_table(1).view(np.uint32), Oracle.osc0().view(np.uint32))
    assert np.array_equal(atsc_amd.get_table(2).view(np.uint32), Oracle.scale().view(np.uint32))
    assert np.array_equal(atsc_amd.get_table(3), Oracle.prbs())
    assert np.array_equal(atsc_amd.get_table(4), Oracle.fsync(0))
    assert np.array_equal(atsc_amd.get_table(5), Oracle.fsync(1))
    assert np.array_equal(atsc_amd.get_table(6), Oracle.rs_gen())
    taps = atsc_amd.get_table(0).view(np.uint32)
    assert all(taps[k] == taps[110 - k] for k in range(111)) and taps[111] == 0  # k_mix_fir relies on this


def test_error_strings_and_arg_checks():
    lib = atsc_amd.load_library()
    assert b"no CPU fallback" in lib.atsc_hip_error_string(-2)
    assert lib.atsc_hip_create(None, None) == -1
    assert lib.atsc_hip_encode_batch(None, None, 1, None, None, None) == -1
    assert lib.atsc_hip_get_table(None, 99, C.c_void_p(1), 4) == -1


def test_no_device_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(atsc_amd.AtscHipError, match="no HIP device"):
        atsc_amd.Encoder(channels=1, max_fields=1)


def test_product_never_touches_the_oracle():
    """The product path must not import/link/call anything under oracle/ (tier rule 3)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "atsc_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("oracle_lib", "libatsc_oracle", "atsc_oracle", "libatsc_ref", "oracle/", "ref_probe"):
                    assert needle not in txt, f"{f} refers to {needle}: the product must not use the checker"
    import subprocess
    out = subprocess.run(["ldd", atsc_amd.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out and "atsc_ref" not in out


def test_headers_compile_as_plain_c(tmp_path):
    """The boundary is a C ABI: every public header must compile as C99 (no C++ types, no HIP headers) and a C program
    must link against the library using only those headers."""
    import subprocess
    src = tmp_path / "use_c.c"
    src.write_text(
        '#include "atsc_hip.h"\n#include "atsc_ts_ingest.h"\n#include "atsc_hip_resample.h"\n#include <stdio.h>\n'
        "int main(void) {\n"
        "    atsc_hip_config cfg = {0, 1, 1, 0};\n"
        "    atsc_hip_encoder* enc = 0;\n"
        "    int rc = atsc_hip_create(&cfg, &enc);\n"
        '    printf("%d %s %u\\n", rc, atsc_hip_error_string(rc), (unsigned)atsc_hip_iq_sample_bytes(ATSC_HIP_IQ_CS16));\n'
        "    if (rc == 0) atsc_hip_destroy(enc);\n"
        "    atsc_ts_ingest* in = atsc_ts_ingest_create(0, 0);\n"
        "    atsc_ts_ingest_destroy(in);\n"
        "    return 0;\n}\n")
    exe = tmp_path / "use_c"
    libdir = os.path.dirname(atsc_amd.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                        "-L", libdir, "-latsc_hip", f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    rc, rest = out.stdout.split(" ", 1)
    assert rest.strip().endswith(" 4")
    import torch
    if not torch.cuda.is_available():
        assert int(rc) != 0 and "no" in rest.lower()       # no device here: a loud error code, not a fallback


def test_reference_caller_builds_unchanged_on_the_adapter():
    """INTEGRATION.md's drop-in claim: the reference's own driver, util/atsc_encoder_null.cpp, compiled UNMODIFIED with
    `atsc/atsc.h` resolving to atsc_amd/host/atsc_encoder.hpp and linked against libatsc_hip.so (oracle/Makefile,
    -Wall -Werror).  Build-container only: /root/reference does not travel; nothing of it is copied into the repo."""
    import subprocess
    if not os.path.isdir("/root/reference/util"):
        pytest.skip("reference tree absent")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "_ref/ref_encoder_null_hip"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_encoder_null_hip")
    out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libatsc_hip.so" in out and "libatsc_ref" not in out
    import torch
    if not torch.cuda.is_available():
        run = subprocess.run([exe, "/dev/null"], capture_output=True, text=True)
        assert run.returncode != 0 and "no HIP device" in run.stderr     # loud, no CPU fallback


def test_fir_error_budget_every_term_from_the_librarys_operand():
    """ADVICE r2 (medium) / VERDICT r2 item 5: every term of the matrix-core forms' error -- tap corrections left out, tap
    residual, z correction left out, z split residual, the Alo.Zlo product never formed, fp32 accumulation along the chain,
    the fp16 underflow floor, max |T_f| -- computed from the tap operand the library builds (atsc_hip_get_table ids 8 / 9:
    the bits create() uploads; tests/test_gpu_round3.py repeats it on the bits read back from the device) and summed:
    below 1e-5 of full scale for k_mix_fir_mm's two phases and for k_mix_fir_w.  scripts/fir_error_budget.py states what
    the accumulation term is charged and why."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fir_error_budget", os.path.join(ROOT, "scripts", "fir_error_budget.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert atsc_amd.get_table(8).size == 2 * 2 * 9 * 64 * 8 and atsc_amd.get_table(9).size == 2 * 8 * 64 * 8
    # the operand is the fp16 split of the shipped taps at the kernels' scales: hi + lo reproduces them to 2^-21
    hi, lo, true = mod.fused_operand()
    assert np.abs(hi + lo - true).max() <= 2.0 ** -21 * np.abs(true).max()
    assert np.abs(true[0]).sum(axis=(0, 2)).max() < 0.19 and (true[0] != 0).sum(axis=(0, 2)).max() == 111
    whi, wlo, wtrue = mod.wide_operand()
    assert np.abs(whi + wlo - wtrue).max() <= 2.0 ** -21 * np.abs(wtrue).max()
    assert abs(mod.osc_max(8) - 0.9) < 1e-6
    rep = mod.report(fields=8)
    assert set(rep) == {"k_mix_fir_mm, symbol phase", "k_mix_fir_mm, half-symbol phase (2x form)", "k_mix_fir_w"}
    for name, (total, terms) in rep.items():
        assert len(terms) >= 4 and all(v >= 0 for v in terms.values())
        assert abs(sum(terms.values()) - total) < 2e-6        # (worst rows of the terms need not coincide)
        assert total < 1e-5, (name, total, terms)
    # what was NOT taken stays outside: two fewer z corrections in k_mix_fir_mm, two fewer low-order products in k_mix_fir_w
    assert mod.fused_budget(0, range(2, 7), range(2, 7), fields=2)[0] > 1e-5
    assert mod.wide_budget(range(2, 6), fields=2)[0] > 1e-5


def test_fir_kernels_do_not_spill(tmp_path):
    """The matrix-core FIR kernels sit at the edge of their register budget (248 of 256 VGPRs at two waves per SIMD).  A
    spill inside the tile loop goes through scratch memory, whose loads share the in-order counter of the IQ stores: the
    wave then waits for its newest stores at every tile and the launch takes 2.7 ms instead of 1.95 (measured with
    side-work placements that differ by one chain slot).  Read the code objects' metadata out of the built library."""
    import re
    import shutil
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(f"{llvm}/llvm-objdump") and os.path.exists(f"{llvm}/llvm-readelf")):
        pytest.skip("llvm-objdump / llvm-readelf not available")
    lib = shutil.copy(atsc_amd.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([f"{llvm}/llvm-objdump", "--offloading", str(lib)], cwd=tmp_path, check=True, capture_output=True)
    seen = 0
    for co in tmp_path.glob("lib.so.*gfx950"):
        notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", str(co)], check=True, capture_output=True, text=True).stdout
        for entry in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", entry)
            if not name or ("k_mix_fir_mm" not in name.group(1) and "k_mix_fir_w" not in name.group(1)):
                continue
            seen += 1
            assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", entry).group(1)) == 0, name.group(1)
            assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", entry).group(1)) == 0, name.group(1)
    assert seen == 13, seen    # 3 sample formats x fused / not fused x 1x / 2x, and the wide form
