"""ctypes binding of the CPU oracle (oracle/libgdg_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (go-dsp-guitar_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgdg_oracle.so")

UNIT_NAMES = [
    "signal_generator", "noise_gate", "bandpass", "auto_wah", "auto_yoy", "compressor", "octaver",
    "excess", "fuzz", "overdrive", "distortion", "tone_stack", "chorus", "flanger", "phaser",
    "tremolo", "ring_modulator", "delay", "reverb", "power_amp", "cabinet",
]
UNIT = {name: i for i, name in enumerate(UNIT_NAMES)}

SCALING_DEFAULT, SCALING_ORTHONORMAL, MODE_STANDARD, MODE_INPLACE = 0, 1, 2, 3


_JITTER_PATH = os.path.join(_HERE, "libgdg_oracle_jitter.so")


def build(force=False, jitter=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile).  jitter: the second build whose transcendental functions are perturbed by a
    seeded +-2 ulp (libm_jitter.h) -- what another libm (Go's math, ocml) may return for the same argument."""
    path, target = (_JITTER_PATH, "libgdg_oracle_jitter.so") if jitter else (_LIB_PATH, "libgdg_oracle.so")
    srcs = [f for f in os.listdir(_HERE) if f.endswith((".c", ".h", ".inc"))]
    newest = max(os.path.getmtime(os.path.join(_HERE, f)) for f in srcs)
    if force or not os.path.exists(path) or os.path.getmtime(path) < newest:
        subprocess.check_call(["make", "-s", "-C", _HERE, target])
    return path


_lib = None
_plain_lib = None
_jitter_lib = None


class jittered:
    """with oracle.jittered(seed): every oracle object made AND used inside runs on the build whose exp / sin / cos / atan / pow / log10 / log2
    results are moved by a seeded -2 .. +2 ulp.  Objects must not cross the boundary (they are destroyed by the library that made them)."""

    def __init__(self, seed=1):
        self.seed = seed

    def __enter__(self):
        global _lib, _plain_lib, _jitter_lib
        lib()
        _plain_lib = _lib
        if _jitter_lib is None:
            build(jitter=True)
            L = C.CDLL(_JITTER_PATH)
            _declare(L)
            L.gdgo_jitter_seed.restype = None
            L.gdgo_jitter_seed.argtypes = [C.c_uint64]
            L.gdgo_jitter_calls.restype = C.c_uint64
            L.gdgo_jitter_calls.argtypes = []
            _jitter_lib = L
        _jitter_lib.gdgo_jitter_seed(self.seed)
        _lib = _jitter_lib
        return self

    def calls(self):
        return int(_jitter_lib.gdgo_jitter_calls())

    def __exit__(self, *exc):
        global _lib
        import gc
        gc.collect()                       # objects of this block die under the library that made them
        _lib = _plain_lib
        return False

_dp = C.POINTER(C.c_double)


class TunerResult(C.Structure):
    _fields_ = [("frequency", C.c_double), ("note_index", C.c_int32), ("cents", C.c_int8)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        _declare(L)
        _lib = L
    return _lib


def _declare(L):
    if True:
        vp = C.c_void_p
        sig = {
            "gdgo_fft_create": (vp, []),
            "gdgo_fft_destroy": (None, [vp]),
            "gdgo_next_power_of_two": (C.c_uint64, [C.c_uint64, C.POINTER(C.c_uint32)]),
            "gdgo_fft_fourier": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int]),
            "gdgo_fft_inverse_fourier": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int]),
            "gdgo_fft_real_fourier": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_int]),
            "gdgo_fft_real_inverse_fourier": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_int]),
            "gdgo_fft_shift": (None, [vp, C.c_int, C.c_int]),
            "gdgo_prng_init": (None, [vp, C.c_uint64]),
            "gdgo_prng_next_float": (C.c_double, [vp]),
            "gdgo_ring_create": (vp, [C.c_int]),
            "gdgo_ring_destroy": (None, [vp]),
            "gdgo_ring_enqueue": (None, [vp, vp, C.c_int]),
            "gdgo_ring_retrieve": (C.c_int, [vp, vp, C.c_int]),
            "gdgo_lanczos_kernel": (C.c_double, [C.c_double, C.c_double]),
            "gdgo_resample_time_length": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32]),
            "gdgo_resample_time": (None, [vp, C.c_int, C.c_uint32, C.c_uint32, vp, C.c_int]),
            "gdgo_resample_frequency": (None, [vp, C.c_int, vp, C.c_uint32]),
            "gdgo_resample_oversample": (None, [vp, C.c_int, vp, C.c_int, C.c_uint32]),
            "gdgo_filter_from_coefficients": (vp, [vp, C.c_int, C.c_uint32, C.c_double]),
            "gdgo_filter_empty": (vp, [C.c_uint32]),
            "gdgo_filter_destroy": (None, [vp]),
            "gdgo_filter_length": (C.c_int, [vp]),
            "gdgo_filter_coefficients": (_dp, [vp]),
            "gdgo_filter_add": (vp, [vp, vp]),
            "gdgo_filter_multiply": (vp, [vp, C.c_double]),
            "gdgo_filter_normalize": (vp, [vp]),
            "gdgo_filter_reduce": (vp, [vp, C.c_uint32]),
            "gdgo_filter_process": (C.c_int, [vp, vp, vp, C.c_int]),
            "gdgo_osd_create": (vp, [C.c_uint32]),
            "gdgo_osd_destroy": (None, [vp]),
            "gdgo_osd_oversample": (C.c_int, [vp, vp, C.c_int, vp, C.c_int]),
            "gdgo_osd_decimate": (C.c_int, [vp, vp, C.c_int, vp, C.c_int]),
            "gdgo_osd_taps": (_dp, [C.c_uint32, C.POINTER(C.c_int)]),
            "gdgo_unit_create": (vp, [C.c_int]),
            "gdgo_unit_destroy": (None, [vp]),
            "gdgo_unit_type": (C.c_int, [vp]),
            "gdgo_unit_param_count": (C.c_int, [vp]),
            "gdgo_unit_set_param": (C.c_int, [vp, C.c_int, C.c_int32]),
            "gdgo_unit_get_param": (C.c_int32, [vp, C.c_int]),
            "gdgo_unit_set_fir": (C.c_int, [vp, vp, C.c_int]),
            "gdgo_unit_process": (None, [vp, vp, vp, C.c_int, C.c_uint32]),
            "gdgo_chain_create": (vp, []),
            "gdgo_chain_destroy": (None, [vp]),
            "gdgo_chain_append_unit": (C.c_int, [vp, C.c_int]),
            "gdgo_chain_remove_unit": (C.c_int, [vp, C.c_int]),
            "gdgo_chain_move_up": (C.c_int, [vp, C.c_int]),
            "gdgo_chain_move_down": (C.c_int, [vp, C.c_int]),
            "gdgo_chain_set_bypass": (C.c_int, [vp, C.c_int, C.c_int]),
            "gdgo_chain_length": (C.c_int, [vp]),
            "gdgo_chain_unit": (vp, [vp, C.c_int]),
            "gdgo_chain_process": (None, [vp, vp, C.c_int, vp, C.c_int, C.c_uint32]),
            "gdgo_tuner_create": (vp, []),
            "gdgo_tuner_destroy": (None, [vp]),
            "gdgo_tuner_process": (None, [vp, vp, C.c_int, C.c_uint32]),
            "gdgo_tuner_analyze": (C.c_int, [vp, C.POINTER(TunerResult)]),
            "gdgo_tuner_note_count": (C.c_int, []),
            "gdgo_tuner_note_name": (C.c_char_p, [C.c_int]),
            "gdgo_tuner_note_frequency": (C.c_double, [C.c_int]),
            "gdgo_tuner_correlation": (_dp, [vp, C.POINTER(C.c_int)]),
            "gdgo_spatializer_create": (vp, [C.c_uint32]),
            "gdgo_spatializer_destroy": (None, [vp]),
            "gdgo_spatializer_set_azimuth": (C.c_int, [vp, C.c_uint32, C.c_double]),
            "gdgo_spatializer_set_distance": (C.c_int, [vp, C.c_uint32, C.c_double]),
            "gdgo_spatializer_set_level": (C.c_int, [vp, C.c_uint32, C.c_double]),
            "gdgo_spatializer_set_sample_rate": (None, [vp, C.c_uint32]),
            "gdgo_spatializer_process": (None, [vp, vp, C.c_int, C.c_int, vp, vp, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- fft -------------------------------------------------------------------------------
def next_power_of_two(v):
    e = C.c_uint32(0)
    p = lib().gdgo_next_power_of_two(v, C.byref(e))
    return p, e.value


def _cplx_buf(z):
    z = np.ascontiguousarray(z, dtype=np.complex128).copy()
    return z


def fourier(z, scaling=SCALING_DEFAULT, mode=MODE_INPLACE):
    z = _cplx_buf(z)
    ft = lib().gdgo_fft_create()
    lib().gdgo_fft_fourier(ft, _ptr(z), len(z), scaling, mode)
    lib().gdgo_fft_destroy(ft)
    return z


def inverse_fourier(z, scaling=SCALING_DEFAULT, mode=MODE_INPLACE):
    z = _cplx_buf(z)
    ft = lib().gdgo_fft_create()
    lib().gdgo_fft_inverse_fourier(ft, _ptr(z), len(z), scaling, mode)
    lib().gdgo_fft_destroy(ft)
    return z


def real_fourier(x, scaling=SCALING_DEFAULT, n_out=None):
    x = _f64(x)
    n_out = len(x) if n_out is None else n_out
    out = np.zeros(n_out, dtype=np.complex128)
    ft = lib().gdgo_fft_create()
    rc = lib().gdgo_fft_real_fourier(ft, _ptr(x), len(x), _ptr(out), n_out, scaling)
    lib().gdgo_fft_destroy(ft)
    return rc, out


def real_inverse_fourier(z, scaling=SCALING_DEFAULT, n_out=None):
    z = _cplx_buf(z)
    n_out = len(z) if n_out is None else n_out
    out = np.zeros(n_out, dtype=np.float64)
    ft = lib().gdgo_fft_create()
    rc = lib().gdgo_fft_real_inverse_fourier(ft, _ptr(z), len(z), _ptr(out), n_out, scaling)
    lib().gdgo_fft_destroy(ft)
    return rc, out


def shift(z, inverse=False):
    z = _cplx_buf(z)
    lib().gdgo_fft_shift(_ptr(z), len(z), 1 if inverse else 0)
    return z


# ---- random / circular -------------------------------------------------------------------
class Prng:
    def __init__(self, seed):
        self._s = (C.c_uint64 * 4)()
        lib().gdgo_prng_init(C.byref(self._s), seed & 0xFFFFFFFFFFFFFFFF)

    def next_float(self):
        return lib().gdgo_prng_next_float(C.byref(self._s))

    def floats(self, n):
        return np.array([self.next_float() for _ in range(n)])


class Ring:
    def __init__(self, size):
        self.n = size
        self._h = lib().gdgo_ring_create(size)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gdgo_ring_destroy(self._h)
            self._h = None

    def enqueue(self, elems):
        e = _f64(elems)
        lib().gdgo_ring_enqueue(self._h, _ptr(e), len(e))

    def retrieve(self, m=None):
        m = self.n if m is None else m
        out = np.zeros(m)
        rc = lib().gdgo_ring_retrieve(self._h, _ptr(out), m)
        return rc, out


# ---- resample ------------------------------------------------------------------------------
def resample_time(x, source_rate, target_rate):
    x = _f64(x)
    n_out = lib().gdgo_resample_time_length(len(x), source_rate, target_rate)
    out = np.zeros(max(n_out, 0))
    lib().gdgo_resample_time(_ptr(x), len(x), source_rate, target_rate, _ptr(out), n_out)
    return out


def resample_frequency(z, n_target):
    z = _cplx_buf(z)
    out = np.zeros(n_target, dtype=np.complex128)
    lib().gdgo_resample_frequency(_ptr(z), len(z), _ptr(out), n_target)
    return out


def resample_oversample(x, n_target, factor):
    x = _f64(x)
    out = np.zeros(n_target)
    lib().gdgo_resample_oversample(_ptr(x), len(x), _ptr(out), n_target, factor)
    return out


# ---- filter --------------------------------------------------------------------------------
class Filter:
    def __init__(self, coeffs=None, sample_rate=0, gain_compensation=0.0, _handle=None):
        if _handle is not None:
            self._h = _handle
        else:
            c = _f64(coeffs if coeffs is not None else [])
            self._h = lib().gdgo_filter_from_coefficients(_ptr(c), len(c), sample_rate, gain_compensation)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gdgo_filter_destroy(self._h)
            self._h = None

    def coefficients(self):
        n = lib().gdgo_filter_length(self._h)
        p = lib().gdgo_filter_coefficients(self._h)
        return np.array([p[i] for i in range(n)]) if n <= 4096 else np.ctypeslib.as_array(p, shape=(n,)).copy()

    def add(self, other):
        h = lib().gdgo_filter_add(self._h, other._h if other is not None else None)
        return Filter(_handle=h) if h else None

    def multiply(self, s):
        return Filter(_handle=lib().gdgo_filter_multiply(self._h, s))

    def normalize(self):
        return Filter(_handle=lib().gdgo_filter_normalize(self._h))

    def reduce(self, order):
        return Filter(_handle=lib().gdgo_filter_reduce(self._h, order))

    def process(self, x):
        x = _f64(x)
        out = np.zeros(len(x))
        rc = lib().gdgo_filter_process(self._h, _ptr(x), _ptr(out), len(x))
        if rc != 0:
            raise ValueError("filter.Process: rc=%d" % rc)
        return out


# ---- oversampling ---------------------------------------------------------------------------
class OversamplerDecimator:
    def __init__(self, factor):
        self.factor = factor
        self._h = lib().gdgo_osd_create(factor)
        if not self._h:
            raise ValueError("unsupported factor")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gdgo_osd_destroy(self._h)
            self._h = None

    def oversample(self, x):
        x = _f64(x)
        out = np.zeros(len(x) * self.factor)
        rc = lib().gdgo_osd_oversample(self._h, _ptr(x), len(x), _ptr(out), len(out))
        assert rc == 0
        return out

    def decimate(self, x):
        x = _f64(x)
        out = np.zeros(len(x) // self.factor)
        rc = lib().gdgo_osd_decimate(self._h, _ptr(x), len(x), _ptr(out), len(out))
        assert rc == 0
        return out


def aa_taps(factor):
    n = C.c_int(0)
    p = lib().gdgo_osd_taps(factor, C.byref(n))
    return np.array([p[i] for i in range(n.value)])


# ---- units / chain --------------------------------------------------------------------------
class Unit:
    def __init__(self, unit_type, _handle=None, _owned=True):
        if isinstance(unit_type, str):
            unit_type = UNIT[unit_type]
        self._owned = _owned
        self._h = _handle if _handle is not None else lib().gdgo_unit_create(unit_type)
        if not self._h:
            raise ValueError("bad unit type")

    def __del__(self):
        if getattr(self, "_h", None) and self._owned:
            lib().gdgo_unit_destroy(self._h)
        self._h = None

    @property
    def type(self):
        return lib().gdgo_unit_type(self._h)

    def set_param(self, idx, value):
        rc = lib().gdgo_unit_set_param(self._h, idx, int(value))
        assert rc == 0, "bad param index"

    def get_param(self, idx):
        return lib().gdgo_unit_get_param(self._h, idx)

    def set_params(self, values):
        for i, v in enumerate(values):
            self.set_param(i, v)

    def set_fir(self, taps):
        t = _f64(taps)
        assert lib().gdgo_unit_set_fir(self._h, _ptr(t), len(t)) == 0

    def process(self, x, sample_rate):
        x = _f64(x)
        out = np.zeros(len(x))
        lib().gdgo_unit_process(self._h, _ptr(x), _ptr(out), len(x), sample_rate)
        return out


class Chain:
    def __init__(self):
        self._h = lib().gdgo_chain_create()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gdgo_chain_destroy(self._h)
            self._h = None

    def append_unit(self, unit_type, bypass=False, params=None, fir=None):
        if isinstance(unit_type, str):
            unit_type = UNIT[unit_type]
        i = lib().gdgo_chain_append_unit(self._h, unit_type)
        if i < 0:
            raise ValueError("bad unit type")
        lib().gdgo_chain_set_bypass(self._h, i, 1 if bypass else 0)
        u = self.unit(i)
        if params is not None:
            u.set_params(params)
        if fir is not None:
            u.set_fir(fir)
        return i

    def unit(self, i):
        h = lib().gdgo_chain_unit(self._h, i)
        if not h:
            raise IndexError(i)
        return Unit(0, _handle=h, _owned=False)

    def remove_unit(self, i):
        return lib().gdgo_chain_remove_unit(self._h, i)

    def move_up(self, i):
        return lib().gdgo_chain_move_up(self._h, i)

    def move_down(self, i):
        return lib().gdgo_chain_move_down(self._h, i)

    def set_bypass(self, i, b):
        return lib().gdgo_chain_set_bypass(self._h, i, 1 if b else 0)

    def length(self):
        return lib().gdgo_chain_length(self._h)

    def process(self, x, sample_rate, n_out=None):
        x = _f64(x)
        n_out = len(x) if n_out is None else n_out
        out = np.zeros(n_out)
        lib().gdgo_chain_process(self._h, _ptr(x), len(x), _ptr(out), n_out, sample_rate)
        return out


# ---- tuner ----------------------------------------------------------------------------------
class Tuner:
    def __init__(self):
        self._h = lib().gdgo_tuner_create()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gdgo_tuner_destroy(self._h)
            self._h = None

    def process(self, samples, sample_rate):
        s = _f64(samples)
        lib().gdgo_tuner_process(self._h, _ptr(s), len(s), sample_rate)

    def analyze(self):
        r = TunerResult()
        rc = lib().gdgo_tuner_analyze(self._h, C.byref(r))
        assert rc == 0
        return {"frequency": r.frequency, "note_index": r.note_index, "cents": r.cents,
                "note": lib().gdgo_tuner_note_name(r.note_index).decode()}

    def correlation(self):
        n = C.c_int(0)
        p = lib().gdgo_tuner_correlation(self._h, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()


def note_table():
    L = lib()
    return [(L.gdgo_tuner_note_name(i).decode(), L.gdgo_tuner_note_frequency(i)) for i in range(L.gdgo_tuner_note_count())]


# ---- spatializer ----------------------------------------------------------------------------
class Spatializer:
    def __init__(self, channels):
        self.channels = channels
        self._h = lib().gdgo_spatializer_create(channels)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gdgo_spatializer_destroy(self._h)
            self._h = None

    def set_azimuth(self, ch, v):
        return lib().gdgo_spatializer_set_azimuth(self._h, ch, v)

    def set_distance(self, ch, v):
        return lib().gdgo_spatializer_set_distance(self._h, ch, v)

    def set_level(self, ch, v):
        return lib().gdgo_spatializer_set_level(self._h, ch, v)

    def set_sample_rate(self, rate):
        lib().gdgo_spatializer_set_sample_rate(self._h, rate)

    def process(self, inputs, aux=None):
        x = _f64(inputs)
        assert x.ndim == 2
        n_in, n = x.shape
        ptrs = (C.c_void_p * n_in)(*[x[i].ctypes.data for i in range(n_in)])
        left, right = np.zeros(n), np.zeros(n)
        a = _f64(aux) if aux is not None else None
        lib().gdgo_spatializer_process(self._h, ptrs, n_in, n, _ptr(a) if a is not None else None, _ptr(left), _ptr(right))
        return left, right


# ---- wave sample codecs / level meter --------------------------------------------------------------
WAVE_FORMATS = {"lpcm8": 0, "lpcm16": 1, "lpcm24": 2, "lpcm32": 3, "ieee32": 4, "ieee64": 5}


class Meter(C.Structure):
    _fields_ = [("enabled", C.c_int), ("current_value", C.c_double), ("peak_value", C.c_double), ("sample_counter", C.c_uint64)]


def _io_lib():
    L = lib()
    if not getattr(L, "_io_ready", False):
        L.gdgo_wave_bytes_per_sample.restype = C.c_int
        L.gdgo_wave_bytes_per_sample.argtypes = [C.c_int]
        L.gdgo_wave_encode.restype = C.c_int
        L.gdgo_wave_encode.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.gdgo_wave_decode.restype = C.c_int
        L.gdgo_wave_decode.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.gdgo_meter_init.argtypes = [C.POINTER(Meter)]
        L.gdgo_meter_set_enabled.argtypes = [C.POINTER(Meter), C.c_int]
        L.gdgo_meter_process.argtypes = [C.POINTER(Meter), C.c_void_p, C.c_size_t, C.c_uint32]
        L.gdgo_meter_analyze.argtypes = [C.POINTER(Meter), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.gdgo_metronome_init.argtypes = [C.c_void_p]
        L.gdgo_metronome_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L._io_ready = True
    return L


def wave_encode(fmt, samples):
    L = _io_lib()
    f = WAVE_FORMATS[fmt] if isinstance(fmt, str) else fmt
    s = _f64(samples)
    out = np.zeros(len(s) * L.gdgo_wave_bytes_per_sample(f), dtype=np.uint8)
    assert L.gdgo_wave_encode(f, _ptr(s), len(s), _ptr(out)) == 0
    return out


def wave_decode(fmt, data):
    L = _io_lib()
    f = WAVE_FORMATS[fmt] if isinstance(fmt, str) else fmt
    d = np.ascontiguousarray(data, dtype=np.uint8)
    n = len(d) // L.gdgo_wave_bytes_per_sample(f)
    out = np.zeros(n)
    assert L.gdgo_wave_decode(f, _ptr(d), n, _ptr(out)) == 0
    return out


class ChannelMeter:
    """level.channelMeterStruct (level/level.go:147-210)"""

    def __init__(self):
        self._m = Meter()
        _io_lib().gdgo_meter_init(C.byref(self._m))

    def set_enabled(self, on):
        _io_lib().gdgo_meter_set_enabled(C.byref(self._m), 1 if on else 0)

    def process(self, buf, sample_rate):
        b = _f64(buf)
        _io_lib().gdgo_meter_process(C.byref(self._m), _ptr(b), len(b), sample_rate)

    def analyze(self):
        lv, pk = C.c_int32(0), C.c_int32(0)
        _io_lib().gdgo_meter_analyze(C.byref(self._m), C.byref(lv), C.byref(pk))
        return lv.value, pk.value

    @property
    def state(self):
        return self._m.current_value, self._m.peak_value, self._m.sample_counter


class MetronomeState(C.Structure):
    _fields_ = [("sample_counter", C.c_uint32), ("tick_counter", C.c_uint32), ("beats_per_period", C.c_uint32),
                ("bpm_speed", C.c_uint32), ("sample_rate", C.c_uint32)]


class Metronome:
    """metronome.metronomeStruct (metronome/metronome.go:63-131)"""

    def __init__(self):
        self.s = MetronomeState()
        _io_lib().gdgo_metronome_init(C.byref(self.s))
        self.tick = self.tock = None

    def process(self, n):
        out = np.zeros(n)
        t = _f64(self.tick) if self.tick is not None else None
        k = _f64(self.tock) if self.tock is not None else None
        _io_lib().gdgo_metronome_process(C.byref(self.s), _ptr(t) if t is not None else None, len(t) if t is not None else 0,
                                         _ptr(k) if k is not None else None, len(k) if k is not None else 0, _ptr(out), n)
        return out
