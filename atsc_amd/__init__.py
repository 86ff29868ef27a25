"""atsc_amd -- MI355X-native ATSC 8VSB modulator hot path.

The product is the C-ABI shared library ``atsc_amd/lib/libatsc_hip.so`` (HIP kernels for gfx950,
sources in ``atsc_amd/csrc``; C++ adapter with the reference's ``atsc::atsc_encoder`` shape in
``atsc_amd/host``).  This Python module is only a thin ctypes binding used by the tests and by
``bench.py``; it contains no compute and has NO CPU fallback: if the library is missing, or no GPU
is present, every entry point raises.
"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("ATSC_HIP_LIB") or os.path.join(ROOT, "atsc_amd", "lib", "libatsc_hip.so")

PACKETS_PER_FIELD = 312
TS_PACKET_BYTES = 188
FIELD_TS_BYTES = 312 * 188
FIELD_CODED_BYTES = 312 * 207
FIELD_SYMBOLS = 313 * 832
FIELD_IQ_BYTES = FIELD_SYMBOLS * 8
PADDED_SYMBOLS = 314 * 832
INTERLEAVER_TAIL = 10608
SYMBOL_RATE = 4500000.0 / 286 * 684  # 10.762238 MHz, one real-time channel

FLAG_TIMING = 1
FLAG_TAPS = 2
FLAG_STRICT_FP32 = 4
MAX_SEEK_STEPS = 1 << 24
(STAGE_RAND_RS, STAGE_INTERLEAVE, STAGE_TRELLIS_PAR, STAGE_TRELLIS_SCAN, STAGE_TRELLIS_MAP, STAGE_OSC, STAGE_MIX_FIR,
 STAGE_STATE, NSTAGES) = range(9)
STAGE_NAMES = ("rand_rs", "interleave", "trellis_parity", "trellis_scan", "trellis_map", "osc_table", "mix_fir",
               "save_state")

# every symbol include/atsc_hip.h declares
EXPORTS = ("atsc_hip_create", "atsc_hip_destroy", "atsc_hip_process", "atsc_hip_encode_batch",
           "atsc_hip_encode_begin", "atsc_hip_encode_finish",
           "atsc_hip_encode_batch_host", "atsc_hip_mix_fir", "atsc_hip_mix_fir_static", "atsc_hip_export_state", "atsc_hip_import_state", "atsc_hip_seek",
           "atsc_hip_span_parity", "atsc_hip_stage_buffer", "atsc_hip_last_timings", "atsc_hip_timing_reserve",
           "atsc_hip_batch_timings", "atsc_hip_timing_select", "atsc_hip_get_table", "atsc_hip_set_iq_format", "atsc_hip_iq_sample_bytes", "atsc_hip_set_fir_form", "atsc_hip_get_fir_form", "atsc_hip_set_upsample", "atsc_hip_get_upsample",
           "atsc_hip_process_chunked", "atsc_hip_process_multi", "atsc_hip_multi_pending",
           "atsc_hip_error_string", "atsc_hip_version",
           # include/atsc_hip_resample.h
           "atsc_hip_resampler_design", "atsc_hip_resampler_create", "atsc_hip_resampler_destroy",
           "atsc_hip_resampler_out_count", "atsc_hip_resample", "atsc_hip_resampler_reset",
           # include/atsc_ts_ingest.h
           "atsc_ts_ingest_create", "atsc_ts_ingest_destroy", "atsc_ts_ingest_push", "atsc_ts_ingest_available",
           "atsc_ts_ingest_pop", "atsc_ts_ingest_pop_padded", "atsc_ts_ingest_get_stats", "atsc_ts_null_packet")


class AtscHipError(RuntimeError):
    code = None   # the C-ABI return code, when the error came from one


class Config(C.Structure):
    _fields_ = [("device", C.c_int), ("channels", C.c_int), ("max_fields", C.c_int), ("flags", C.c_int)]


class StreamState(C.Structure):
    _fields_ = [("field_index", C.c_uint64), ("coded_tail", C.c_uint8 * INTERLEAVER_TAIL),
                ("last_group", C.c_uint8 * 12), ("z2", C.c_uint8 * 12), ("z0_next", C.c_uint8 * 12),
                ("z0_prev", C.c_uint8 * 12)]


class IngestStats(C.Structure):
    _fields_ = [("bytes_in", C.c_uint64), ("bytes_dropped", C.c_uint64), ("packets_out", C.c_uint64),
                ("nulls_stuffed", C.c_uint64), ("sync_errors", C.c_uint64), ("sync_losses", C.c_uint64),
                ("locked", C.c_int)]


FIELD_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_uint)
CHUNK_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)
IQ_CF32, IQ_CS16, IQ_CS8 = 0, 1, 2
FIR_AUTO, FIR_FFT, FIR_MFMA, FIR_FUSED, FIR_WIDE = 0, 1, 2, 3, 4
ERR_ARG, ERR_NO_DEVICE, ERR_TOO_LARGE, ERR_STATE = -1, -2, -3, -4   # include/atsc_hip.h
MULTI_MAX_AHEAD = 8   # ATSC_HIP_MULTI_MAX_AHEAD

_lib = None


def build(verbose=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "atsc_amd", "csrc"), "all"], capture_output=True, text=True)
    if verbose or out.returncode:
        print(out.stdout, out.stderr)
    if out.returncode:
        raise AtscHipError("building libatsc_hip.so failed")
    return LIB_PATH


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AtscHipError(f"{LIB_PATH} is missing: run __graft_entry__.build() (there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, u8p = C.c_void_p, C.c_void_p
    L.atsc_hip_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.atsc_hip_destroy.argtypes = [vp]
    L.atsc_hip_destroy.restype = None
    L.atsc_hip_process.argtypes = [vp, u8p, C.c_uint, FIELD_CB, vp]
    L.atsc_hip_encode_batch.argtypes = [vp, u8p, C.c_int, u8p, vp, vp]
    L.atsc_hip_encode_begin.argtypes = [vp, u8p, C.c_int, vp, vp]
    L.atsc_hip_encode_finish.argtypes = [vp, vp, u8p, vp, vp]
    L.atsc_hip_encode_batch_host.argtypes = [vp, u8p, C.c_int, u8p, vp]
    L.atsc_hip_mix_fir.argtypes = [vp, u8p, C.c_int, vp, vp]
    L.atsc_hip_mix_fir_static.argtypes = [C.c_int, u8p, C.c_int, C.c_int, C.c_uint64, vp, C.c_int, vp]
    L.atsc_hip_export_state.argtypes = [vp, C.c_int, C.POINTER(StreamState)]
    L.atsc_hip_import_state.argtypes = [vp, C.c_int, C.POINTER(StreamState)]
    L.atsc_hip_seek.argtypes = [vp, C.c_uint64, vp]
    L.atsc_hip_span_parity.argtypes = [vp, C.c_int, u8p]
    L.atsc_hip_stage_buffer.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.atsc_hip_last_timings.argtypes = [vp, C.POINTER(C.c_float)]
    L.atsc_hip_timing_reserve.argtypes = [vp, C.c_int]
    L.atsc_hip_batch_timings.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.atsc_hip_get_table.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.atsc_hip_set_iq_format.argtypes = [vp, C.c_int]
    L.atsc_hip_set_fir_form.argtypes = [vp, C.c_int]
    L.atsc_hip_get_fir_form.argtypes = [vp]
    L.atsc_hip_set_upsample.argtypes = [vp, C.c_int]
    L.atsc_hip_get_upsample.argtypes = [vp]
    L.atsc_hip_iq_sample_bytes.argtypes = [C.c_int]
    L.atsc_hip_iq_sample_bytes.restype = C.c_size_t
    L.atsc_hip_process_chunked.argtypes = [vp, u8p, C.c_uint, C.c_size_t, CHUNK_CB, vp]
    L.atsc_hip_process_multi.argtypes = [vp, C.c_int, u8p, C.c_uint, FIELD_CB, vp]
    L.atsc_hip_multi_pending.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t)]
    L.atsc_hip_timing_select.argtypes = [vp, C.c_uint]
    L.atsc_hip_error_string.argtypes = [C.c_int]
    L.atsc_hip_error_string.restype = C.c_char_p
    L.atsc_hip_version.restype = C.c_char_p
    L.atsc_hip_resampler_design.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, vp]
    L.atsc_hip_resampler_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(vp)]
    L.atsc_hip_resampler_destroy.argtypes = [vp]
    L.atsc_hip_resampler_destroy.restype = None
    L.atsc_hip_resampler_out_count.argtypes = [vp, C.c_size_t]
    L.atsc_hip_resampler_out_count.restype = C.c_size_t
    L.atsc_hip_resample.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, vp]
    L.atsc_hip_resampler_reset.argtypes = [vp]
    L.atsc_ts_ingest_create.argtypes = [C.c_int, C.c_int]
    L.atsc_ts_ingest_create.restype = vp
    L.atsc_ts_ingest_destroy.argtypes = [vp]
    L.atsc_ts_ingest_destroy.restype = None
    for f in (L.atsc_ts_ingest_push, L.atsc_ts_ingest_pop, L.atsc_ts_ingest_pop_padded):
        f.argtypes = [vp, u8p, C.c_size_t]
        f.restype = C.c_size_t
    L.atsc_ts_ingest_available.argtypes = [vp]
    L.atsc_ts_ingest_available.restype = C.c_size_t
    L.atsc_ts_ingest_get_stats.argtypes = [vp, C.POINTER(IngestStats)]
    L.atsc_ts_ingest_get_stats.restype = None
    L.atsc_ts_null_packet.argtypes = [u8p]
    L.atsc_ts_null_packet.restype = None
    _lib = L
    return L


def check(rc):
    if rc != 0:
        err = AtscHipError(f"atsc_hip error {rc}: {load_library().atsc_hip_error_string(rc).decode()}")
        err.code = rc
        raise err


class Encoder:
    """Host mirror of ``atsc::atsc_encoder`` (reference: src/atsc/atsc.h:7-16) plus the batched entry points.

    ``process(pkts, count, fn)`` has the reference's semantics; ``encode_batch`` takes device pointers
    (ints, e.g. ``tensor.data_ptr()``) and is asynchronous on ``stream``.
    """

    def __init__(self, channels=1, max_fields=1, device=0, timing=False, taps=False, strict_fp32=False):
        self.L = load_library()
        self.channels, self.max_fields, self.device = channels, max_fields, device
        cfg = Config(device, channels, max_fields, (FLAG_TIMING if timing else 0) | (FLAG_TAPS if taps else 0)
                     | (FLAG_STRICT_FP32 if strict_fp32 else 0))
        h = C.c_void_p()
        check(self.L.atsc_hip_create(C.byref(cfg), C.byref(h)))
        self.h = h

    @classmethod
    def create(cls, **kw):
        return cls(**kw)

    def close(self):
        if getattr(self, "h", None):
            self.L.atsc_hip_destroy(self.h)
            self.h = None

    __del__ = close

    # ---- reference-shaped streaming path
    def process(self, pkts, count, fn):
        """pkts: bytes-like / numpy u8 of count*188 bytes; fn(iq_ptr:int, nbytes:int) per completed field."""
        import numpy as np
        buf = np.ascontiguousarray(np.frombuffer(pkts, dtype=np.uint8) if not isinstance(pkts, np.ndarray) else pkts)
        assert buf.size >= count * TS_PACKET_BYTES
        cb = FIELD_CB(lambda user, iq, nbytes: fn(iq, nbytes))
        check(self.L.atsc_hip_process(self.h, buf.ctypes.data, count, cb, None))

    def process_multi(self, channel, pkts, count, fn):
        """atsc_hip_process_multi: stream `channel` of a K-channel handle; fn(iq_ptr:int, nbytes:int) per completed field of
        THAT stream, possibly during another stream's call (the wrapper keeps each stream's latest callback alive)."""
        import numpy as np
        buf = np.ascontiguousarray(np.frombuffer(pkts, dtype=np.uint8) if not isinstance(pkts, np.ndarray) else pkts)
        assert buf.size >= count * TS_PACKET_BYTES
        cb = FIELD_CB(lambda user, iq, nbytes: fn(iq, nbytes))
        if not hasattr(self, "_multi_cbs"):
            self._multi_cbs = {}
        old = self._multi_cbs.get(channel)
        self._multi_cbs[channel] = cb
        try:
            check(self.L.atsc_hip_process_multi(self.h, channel, buf.ctypes.data, count, cb, None))
        finally:
            del old   # released only after the call: the library may have fired it before taking the new one

    def multi_pending(self, channel):
        n = C.c_size_t(0)
        check(self.L.atsc_hip_multi_pending(self.h, channel, C.byref(n)))
        return n.value

    def process_chunked(self, pkts, count, mtu_samples, fn):
        """SDR-sink shape (util/atsc_encoder.cpp:84-100): fn(ptr:int, nsamples:int, fmt:int) per <= mtu piece of a field."""
        import numpy as np
        buf = np.ascontiguousarray(np.frombuffer(pkts, dtype=np.uint8) if not isinstance(pkts, np.ndarray) else pkts)
        assert buf.size >= count * TS_PACKET_BYTES
        cb = CHUNK_CB(lambda user, p, n, fmt: fn(p, n, fmt))
        check(self.L.atsc_hip_process_chunked(self.h, buf.ctypes.data, count, mtu_samples, cb, None))

    def set_iq_format(self, fmt):
        check(self.L.atsc_hip_set_iq_format(self.h, fmt))

    def set_fir_form(self, form):
        check(self.L.atsc_hip_set_fir_form(self.h, form))

    def fir_form(self):
        """The FIR form the next batch runs (AUTO resolved)."""
        form = self.L.atsc_hip_get_fir_form(self.h)
        if form < 0:
            check(form)
        return form

    def set_upsample(self, factor):
        """1 (default) or 2: output samples per symbol (include/atsc_hip.h, atsc_hip_set_upsample)."""
        check(self.L.atsc_hip_set_upsample(self.h, factor))

    def upsample(self):
        return self.L.atsc_hip_get_upsample(self.h)

    # ---- batched device path
    def encode_batch(self, d_ts, fields, d_sym=None, d_iq=None, stream=None):
        check(self.L.atsc_hip_encode_batch(self.h, d_ts, fields, d_sym, d_iq, stream))

    def encode_begin(self, d_ts, fields, d_span_parity=None, stream=None):
        check(self.L.atsc_hip_encode_begin(self.h, d_ts, fields, d_span_parity, stream))

    def encode_finish(self, d_carry_xor=None, d_sym=None, d_iq=None, stream=None):
        check(self.L.atsc_hip_encode_finish(self.h, d_carry_xor, d_sym, d_iq, stream))

    def mix_fir(self, d_sym, fields, d_iq, stream=None):
        check(self.L.atsc_hip_mix_fir(self.h, d_sym, fields, d_iq, stream))

    def encode_batch_host(self, ts, fields, want_sym=True, want_iq=True):
        import numpy as np
        ts = np.ascontiguousarray(ts, np.uint8)
        assert ts.size == self.channels * fields * FIELD_TS_BYTES
        sym = np.empty(self.channels * fields * FIELD_SYMBOLS, np.uint8) if want_sym else None
        iq = np.empty(self.channels * fields * FIELD_SYMBOLS * self.upsample(), np.complex64) if want_iq else None
        check(self.L.atsc_hip_encode_batch_host(self.h, ts.ctypes.data, fields, sym.ctypes.data if want_sym else None,
                                                iq.ctypes.data if want_iq else None))
        return sym, iq

    # ---- state, taps, timings
    def export_state(self, channel=0):
        st = StreamState()
        check(self.L.atsc_hip_export_state(self.h, channel, C.byref(st)))
        return st

    def import_state(self, st, channel=0):
        check(self.L.atsc_hip_import_state(self.h, channel, C.byref(st)))

    def seek(self, field_index, stream=None):
        """atsc_hip_seek: every channel starts at absolute field `field_index` with zero trellis state, in stream order."""
        check(self.L.atsc_hip_seek(self.h, field_index, stream))

    def span_parity(self, channel=0):
        import numpy as np
        out = np.zeros(36, np.uint8)
        check(self.L.atsc_hip_span_parity(self.h, channel, out.ctypes.data))
        return out.reshape(3, 12)

    def stage_buffer(self, stage):
        p, n, st = C.c_void_p(), C.c_size_t(), C.c_size_t()
        check(self.L.atsc_hip_stage_buffer(self.h, stage, C.byref(p), C.byref(n), C.byref(st)))
        return p.value, n.value, st.value

    def last_timings(self):
        ms = (C.c_float * NSTAGES)()
        check(self.L.atsc_hip_last_timings(self.h, ms))
        return {STAGE_NAMES[i]: ms[i] for i in range(NSTAGES)}

    def timing_reserve(self, nslots):
        check(self.L.atsc_hip_timing_reserve(self.h, nslots))

    def timing_select(self, stages=None):
        """atsc_hip_timing_select: time only the named stages (None: all)."""
        mask = 0xFFFFFFFF if stages is None else sum(1 << STAGE_NAMES.index(k) for k in stages)
        check(self.L.atsc_hip_timing_select(self.h, mask))

    def batch_timings(self, batch):
        ms = (C.c_float * NSTAGES)()
        check(self.L.atsc_hip_batch_timings(self.h, batch, ms))
        return {STAGE_NAMES[i]: ms[i] for i in range(NSTAGES)}

    def table(self, which):
        return get_table(which, self.h)


STATIC_MAX_FIELDS = 4


def mix_fir_static(device, d_sym, channels, fields, field_index, d_iq, iq_format=IQ_CF32, stream=None):
    """atsc_hip_mix_fir_static: mixer + FIR with no handle and no allocation (include/atsc_hip.h)."""
    check(load_library().atsc_hip_mix_fir_static(device, d_sym, channels, fields, field_index, d_iq, iq_format, stream))


def get_table(which, handle=None):
    """Constant tables as the library builds them (no GPU needed when handle is None)."""
    import numpy as np
    shapes = {0: (112, np.float32), 1: (PADDED_SYMBOLS * 2, np.float32), 2: (2, np.float32),
              3: (187 * 312, np.uint8), 4: (820, np.uint8), 5: (820, np.uint8), 6: (21, np.uint8), 7: (112, np.float32),
              8: (2 * 2 * 9 * 64 * 8, np.uint16), 9: (2 * 8 * 64 * 8, np.uint16)}
    n, dt = shapes[which]
    out = np.zeros(n, dt)
    check(load_library().atsc_hip_get_table(handle, which, out.ctypes.data, out.nbytes))
    return out


class TsIngest:
    """ctypes view of include/atsc_ts_ingest.h: sync acquisition / resync / null stuffing in front of the encoder."""

    def __init__(self, lock_packets=0, loss_packets=0):
        self.L = load_library()
        self.h = self.L.atsc_ts_ingest_create(lock_packets, loss_packets)
        if not self.h:
            raise AtscHipError("atsc_ts_ingest_create: invalid argument")

    def close(self):
        if getattr(self, "h", None):
            self.L.atsc_ts_ingest_destroy(self.h)
            self.h = None

    __del__ = close

    def push(self, data):
        import numpy as np
        buf = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
        return self.L.atsc_ts_ingest_push(self.h, buf.ctypes.data, buf.size)

    def available(self):
        return self.L.atsc_ts_ingest_available(self.h)

    def pop(self, max_packets):
        import numpy as np
        out = np.empty(max_packets * TS_PACKET_BYTES, np.uint8)
        n = self.L.atsc_ts_ingest_pop(self.h, out.ctypes.data, max_packets)
        return out[:n * TS_PACKET_BYTES]

    def pop_padded(self, packets):
        import numpy as np
        out = np.empty(packets * TS_PACKET_BYTES, np.uint8)
        stuffed = self.L.atsc_ts_ingest_pop_padded(self.h, out.ctypes.data, packets)
        return out, stuffed

    def stats(self):
        st = IngestStats()
        self.L.atsc_ts_ingest_get_stats(self.h, C.byref(st))
        return {k: getattr(st, k) for k, _ in IngestStats._fields_}


def resampler_design(L, M, K=16, beta=0.0):
    """Prototype taps h[0..K*L) as the library designs them (host only)."""
    import numpy as np
    h = np.zeros(K * L, np.float32)
    check(load_library().atsc_hip_resampler_design(L, M, K, beta, h.ctypes.data))
    return h


class Resampler:
    """ctypes view of include/atsc_hip_resample.h (SDR-rate polyphase resampler, L/M times the symbol rate)."""

    def __init__(self, channels, L, M, K=16, beta=0.0, out_format=IQ_CF32, device=0):
        self.L_ = load_library()
        self.channels, self.L, self.M, self.K, self.fmt = channels, L, M, K, out_format
        h = C.c_void_p()
        check(self.L_.atsc_hip_resampler_create(device, channels, L, M, K, beta, out_format, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L_.atsc_hip_resampler_destroy(self.h)
            self.h = None

    __del__ = close

    def out_count(self, n_in):
        return self.L_.atsc_hip_resampler_out_count(self.h, n_in)

    def resample(self, d_in, n_in, in_stride, d_out, out_stride, stream=None):
        check(self.L_.atsc_hip_resample(self.h, d_in, n_in, in_stride, d_out, out_stride, stream))

    def reset(self):
        check(self.L_.atsc_hip_resampler_reset(self.h))
