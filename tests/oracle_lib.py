"""ctypes access to the CPU checkers (TEST INFRASTRUCTURE ONLY).

* ``Oracle``  -- oracle/_build/libatsc_oracle.so, the C restatement (oracle/atsc_oracle.c).
* ``Ref``     -- oracle/_ref/libatsc_ref.so, the compiled reference behind oracle/ref_probe.cpp
                 (present only when it was built in a container that has /root/reference).

Nothing under atsc_amd/ imports this module.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "libatsc_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libatsc_ref.so")

PKTS = 312
FIELD_TS = 312 * 188
FIELD_ENC = 312 * 207
FIELD_SYMS = 313 * 832
PAD_SYMS = FIELD_SYMS + 832

_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


# ----------------------------------------------------------------------------- input generators
def ts_null(nfields):
    """`null` input of SURVEY.md 8(c): packets 47 1F FF 10 + 184 x FF."""
    pkt = np.full(188, 0xFF, np.uint8)
    pkt[:4] = [0x47, 0x1F, 0xFF, 0x10]
    return np.tile(pkt, nfields * PKTS)


def ts_mt(seed, nfields):
    """`mt1234`-style input: low byte of std::mt19937(seed) outputs, every 188th byte = 0x47."""
    bg = np.random.MT19937()
    bg._legacy_seeding(int(seed))
    raw = bg.random_raw(nfields * FIELD_TS)
    b = (raw & 0xFF).astype(np.uint8)
    b[::188] = 0x47
    return b


def ts_mix(seed, nfields, p_null=0.2):
    """`mix` input of SURVEY.md 8(d) config 2: null packet with probability p_null, else PRBS payload."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = nfields * PKTS
    out = rng.integers(0, 256, size=(n, 188), dtype=np.uint8)
    out[:, 0] = 0x47
    nullpkt = ts_null(1)[:188]
    mask = rng.random(n) < p_null
    out[mask] = nullpkt
    return out.reshape(-1)


# ----------------------------------------------------------------------------- C restatement
class Oracle:
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = C.CDLL(ORACLE_SO)
            L.oracle_create.restype = C.c_void_p
            L.oracle_destroy.argtypes = [C.c_void_p]
            L.oracle_process_fields.argtypes = [C.c_void_p, _u8p, C.c_int, _u8p, _u8p, _u8p, _f32p]
            L.oracle_symbols.argtypes = [C.c_void_p, _u8p, C.c_int, _u8p]
            L.oracle_mix_fir.argtypes = [C.c_void_p, _u8p, C.c_int, _f32p]
            L.oracle_get_state.argtypes = [C.c_void_p, _u8p, _u8p, _u8p, _u8p, C.POINTER(C.c_uint64)]
            L.oracle_il_to_ref_layout.argtypes = [_u8p, _u8p]
            L.oracle_rs_parity.argtypes = [_u8p, _u8p]
            for name, t in (("oracle_prbs", _u8p), ("oracle_rs_gen", _u8p), ("oracle_taps", _f32p),
                            ("oracle_osc0", _f32p), ("oracle_scale", _f32p), ("oracle_osc_table", _f32p)):
                getattr(L, name).restype = t
            L.oracle_osc_table.argtypes = [C.c_void_p]
            L.oracle_fsync.restype = _u8p
            L.oracle_fsync.argtypes = [C.c_int]
            L.oracle_init()
            cls._lib = L
        return cls._lib

    def __init__(self):
        self.L = self.lib()
        self.h = C.c_void_p(self.L.oracle_create())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_destroy(self.h)
            self.h = None

    def process_fields(self, ts, nfields, want=("enc", "il", "sym", "iq")):
        ts = np.ascontiguousarray(ts, np.uint8)
        assert ts.size == nfields * FIELD_TS
        out = {}
        out["enc"] = np.empty(nfields * FIELD_ENC, np.uint8) if "enc" in want else None
        out["il"] = np.empty(nfields * FIELD_ENC, np.uint8) if "il" in want else None
        out["sym"] = np.empty(nfields * FIELD_SYMS, np.uint8) if "sym" in want else None
        out["iq"] = np.empty(nfields * FIELD_SYMS * 2, np.float32) if "iq" in want else None
        self.L.oracle_process_fields(self.h, _p(ts, _u8p), nfields, _p(out["enc"], _u8p), _p(out["il"], _u8p),
                                     _p(out["sym"], _u8p), _p(out["iq"], _f32p))
        return {k: v for k, v in out.items() if v is not None}

    def symbols(self, ts, nfields):
        ts = np.ascontiguousarray(ts, np.uint8)
        sym = np.empty(nfields * FIELD_SYMS, np.uint8)
        self.L.oracle_symbols(self.h, _p(ts, _u8p), nfields, _p(sym, _u8p))
        return sym

    def mix_fir(self, sym, nfields):
        sym = np.ascontiguousarray(sym, np.uint8)
        iq = np.empty(nfields * FIELD_SYMS * 2, np.float32)
        self.L.oracle_mix_fir(self.h, _p(sym, _u8p), nfields, _p(iq, _f32p))
        return iq

    def state(self):
        z2, d1, d2, rsv = (np.zeros(12, np.uint8) for _ in range(4))
        fi = C.c_uint64(0)
        self.L.oracle_get_state(self.h, _p(z2, _u8p), _p(d1, _u8p), _p(d2, _u8p), _p(rsv, _u8p), C.byref(fi))
        return dict(z2=z2, d1=d1, d2=d2, reserved=rsv, field_index=fi.value)

    def osc_table(self):
        return np.ctypeslib.as_array(self.L.oracle_osc_table(self.h), shape=(PAD_SYMS * 2,)).copy()

    # static tables
    @classmethod
    def prbs(cls):
        return np.ctypeslib.as_array(cls.lib().oracle_prbs(), shape=(187 * 312,)).copy()

    @classmethod
    def rs_gen(cls):
        return np.ctypeslib.as_array(cls.lib().oracle_rs_gen(), shape=(21,)).copy()

    @classmethod
    def fsync(cls, odd):
        return np.ctypeslib.as_array(cls.lib().oracle_fsync(int(odd)), shape=(820,)).copy()

    @classmethod
    def taps(cls):
        return np.ctypeslib.as_array(cls.lib().oracle_taps(), shape=(112,)).copy()

    @classmethod
    def osc0(cls):
        return np.ctypeslib.as_array(cls.lib().oracle_osc0(), shape=(PAD_SYMS * 2,)).copy()

    @classmethod
    def scale(cls):
        return np.ctypeslib.as_array(cls.lib().oracle_scale(), shape=(2,)).copy()

    @classmethod
    def il_to_ref_layout(cls, il):
        il = np.ascontiguousarray(il, np.uint8)
        out = np.empty_like(il)
        for f in range(il.size // FIELD_ENC):
            a = il[f * FIELD_ENC:(f + 1) * FIELD_ENC]
            b = out[f * FIELD_ENC:(f + 1) * FIELD_ENC]
            cls.lib().oracle_il_to_ref_layout(_p(a, _u8p), _p(b, _u8p))
        return out


# ----------------------------------------------------------------------------- compiled reference
def ref_available():
    return os.path.exists(REF_SO)


class Ref:
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = C.CDLL(REF_SO)
            L.ref_create.restype = C.c_void_p
            L.ref_destroy.argtypes = [C.c_void_p]
            L.ref_process.argtypes = [C.c_void_p, _u8p, C.c_uint, _f32p, C.c_int]
            L.ref_process.restype = C.c_int
            L.ref_process_fields.argtypes = [C.c_void_p, _u8p, C.c_int, _u8p, _u8p, _u8p, _f32p, _f32p]
            L.ref_get_carry.argtypes = [C.c_void_p, _u8p]
            L.ref_osc_table.argtypes = [C.c_void_p, _f32p]
            L.ref_osc_scale.argtypes = [_f32p]
            L.ref_taps.argtypes = [_f32p]
            L.ref_prbs.argtypes = [_u8p]
            L.ref_interleaver_table.argtypes = [C.POINTER(C.c_uint32)]
            cls._lib = L
        return cls._lib

    def __init__(self):
        self.L = self.lib()
        self.h = C.c_void_p(self.L.ref_create())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ref_destroy(self.h)
            self.h = None

    def process(self, pkts, count, max_fields):
        pkts = np.ascontiguousarray(pkts, np.uint8)
        iq = np.empty(max_fields * FIELD_SYMS * 2, np.float32)
        n = self.L.ref_process(self.h, _p(pkts, _u8p), count, _p(iq, _f32p), max_fields)
        return n, iq[:min(n, max_fields) * FIELD_SYMS * 2]

    def process_fields(self, ts, nfields, want=("enc", "il", "sym", "iq")):
        ts = np.ascontiguousarray(ts, np.uint8)
        out = {}
        out["enc"] = np.empty(nfields * FIELD_ENC, np.uint8) if "enc" in want else None
        out["il"] = np.empty(nfields * FIELD_ENC, np.uint8) if "il" in want else None
        out["sym"] = np.empty(nfields * FIELD_SYMS, np.uint8) if "sym" in want else None
        out["mix"] = np.empty(nfields * PAD_SYMS * 2, np.float32) if "mix" in want else None
        out["iq"] = np.empty(nfields * FIELD_SYMS * 2, np.float32) if "iq" in want else None
        self.L.ref_process_fields(self.h, _p(ts, _u8p), nfields, _p(out["enc"], _u8p), _p(out["il"], _u8p),
                                  _p(out["sym"], _u8p), _p(out["mix"], _f32p), _p(out["iq"], _f32p))
        return {k: v for k, v in out.items() if v is not None}

    def carry(self):
        c = np.zeros(36, np.uint8)
        self.L.ref_get_carry(self.h, _p(c, _u8p))
        return c.reshape(3, 12)

    def osc_table(self):
        t = np.empty(PAD_SYMS * 2, np.float32)
        self.L.ref_osc_table(self.h, _p(t, _f32p))
        return t

    @classmethod
    def scale(cls):
        s = np.zeros(2, np.float32)
        cls.lib().ref_osc_scale(_p(s, _f32p))
        return s

    @classmethod
    def taps(cls):
        t = np.zeros(112, np.float32)
        cls.lib().ref_taps(_p(t, _f32p))
        return t

    @classmethod
    def prbs(cls):
        t = np.zeros(187 * 312, np.uint8)
        cls.lib().ref_prbs(_p(t, _u8p))
        return t
