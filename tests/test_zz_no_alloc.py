"""The allocation-free entry point atsc_hip_mix_fir_static (include/atsc_hip.h) against the existing entry point it
is a variant of, atsc_hip_mix_fir: same kernels, so the IQ must be bit-identical.  Two calls in a row, the second with
a smaller input (all calls on a GPU share one device array, which keeps the last call's contents)."""
import numpy as np
import pytest

import atsc_amd
from atsc_amd import FIELD_SYMBOLS

pytestmark = pytest.mark.gpu


def _handle_result(torch, sym, channels, fields, field_index, fmt, sample_bytes):
    enc = atsc_amd.Encoder(channels=channels, max_fields=fields)
    enc.set_fir_form(atsc_amd.FIR_FFT)
    enc.set_iq_format(fmt)
    if field_index:
        for c in range(channels):
            st = enc.export_state(c)
            st.field_index = field_index
            enc.import_state(st, c)
    out = torch.zeros(channels * fields * FIELD_SYMBOLS * sample_bytes, dtype=torch.uint8, device="cuda")
    enc.mix_fir(sym.data_ptr(), fields, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out


def test_mix_fir_static_equals_mix_fir():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    L = atsc_amd.load_library()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda")
    g.manual_seed(99)
    # set-up only (what a caller does once outside graph capture)
    atsc_amd.mix_fir_static(0, None, 1, 0, 0, None)
    cases = [(3, 4, 0, atsc_amd.IQ_CF32, 8),      # first call: 3 channels x 4 fields (the maximum)
             (2, 1, 5, atsc_amd.IQ_CF32, 8),      # second call: smaller, at a later field index
             (1, 2, 1, atsc_amd.IQ_CS16, 4)]
    for channels, fields, fidx, fmt, sb in cases:
        sym = torch.randint(0, 8, (channels * fields * FIELD_SYMBOLS,), dtype=torch.uint8, device="cuda", generator=g)
        guard = 4096
        out = torch.full((channels * fields * FIELD_SYMBOLS * sb + guard,), 0x5A, dtype=torch.uint8, device="cuda")
        # the call allocates nothing: under stream capture a hipMalloc (or a lazy module load) would invalidate the capture,
        # and the device's free memory is the same before and after
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            graph.capture_begin()
            atsc_amd.mix_fir_static(0, sym.data_ptr(), channels, fields, fidx, out.data_ptr(), fmt, side.cuda_stream)
            graph.capture_end()
        graph.replay()
        torch.cuda.synchronize()
        assert bool((out[:64] != 0x5A).any()), "the captured call did not run"
        out.fill_(0x5A)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        atsc_amd.mix_fir_static(0, sym.data_ptr(), channels, fields, fidx, out.data_ptr(), fmt, st)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] >= free0, "device memory shrank across the call"
        assert bool((out[-guard:] == 0x5A).all()), "store past the end of the output"
        want = _handle_result(torch, sym, channels, fields, fidx, fmt, sb)
        assert bool((out[:-guard] == want).all()), f"{channels} ch x {fields} fields at field {fidx}: differs from atsc_hip_mix_fir"
    # sizes beyond the fixed array are refused with a clear error, nothing is launched
    sym = torch.zeros(5 * FIELD_SYMBOLS, dtype=torch.uint8, device="cuda")
    out = torch.zeros(5 * FIELD_SYMBOLS * 8, dtype=torch.uint8, device="cuda")
    assert L.atsc_hip_mix_fir_static(0, sym.data_ptr(), 1, atsc_amd.STATIC_MAX_FIELDS + 1, 0, out.data_ptr(), 0, None) == -3
    assert b"larger" in L.atsc_hip_error_string(-3)
    assert L.atsc_hip_mix_fir_static(0, None, 1, 1, 0, out.data_ptr(), 0, None) == -1
    assert L.atsc_hip_mix_fir_static(99, sym.data_ptr(), 1, 1, 0, out.data_ptr(), 0, None) == -1
    # the caller's current device is left as it was
    assert torch.cuda.current_device() == 0
