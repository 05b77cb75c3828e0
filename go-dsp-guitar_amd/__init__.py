"""go-dsp-guitar's batch-mode effects pipeline on MI355X: Python plumbing over libgdg.so.

The product is the C-ABI shared library (include/gdg.h, csrc/*.hip); this module only loads
it with ctypes and offers thin conveniences for tests and bench.py.  There is no CPU compute
path here: if the library or a GPU is missing, calls fail loudly.

The directory name contains a hyphen (it mirrors the reference's repo name), so import it
through __graft_entry__.load_package() which registers it as `go_dsp_guitar_amd`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgdg.so")
CSRC = os.path.join(_HERE, "csrc")

GDG_OK, GDG_ERR_INVALID, GDG_ERR_UNSUPPORTED, GDG_ERR_HIP, GDG_ERR_NO_DEVICE, GDG_ERR_NOMEM = 0, -1, -2, -3, -4, -5

UNIT_NAMES = [
    "signal_generator", "noise_gate", "bandpass", "auto_wah", "auto_yoy", "compressor", "octaver",
    "excess", "fuzz", "overdrive", "distortion", "tone_stack", "chorus", "flanger", "phaser",
    "tremolo", "ring_modulator", "delay", "reverb", "power_amp", "cabinet",
]
UNIT = {name: i for i, name in enumerate(UNIT_NAMES)}

K_FIR_FWD, K_FIR_MAC, K_FIR_INV, K_SEGMENT, K_TUNER, K_SPATIALIZER, K_WAVE, K_RESAMPLE, K_METER, K_FIR_MAC_CHAIN = range(10)
WAVE_FORMATS = {"lpcm8": 0, "lpcm16": 1, "lpcm24": 2, "lpcm32": 3, "ieee32": 4, "ieee64": 5}     # enum gdg_wave_format
KERNEL_KINDS = ["fir_fwd", "fir_mac", "fir_inv", "segment", "tuner", "spatializer"]

# every symbol include/gdg.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "gdg_version", "gdg_device_count", "gdg_ctx_create", "gdg_ctx_destroy", "gdg_last_error", "gdg_ctx_channels",
    "gdg_ctx_stream", "gdg_ctx_synchronize", "gdg_ctx_share_ir_spectra", "gdg_unit_create", "gdg_unit_destroy", "gdg_unit_set_param",
    "gdg_unit_get_param", "gdg_unit_set_fir", "gdg_unit_compile_fir", "gdg_unit_get_fir", "gdg_unit_reset", "gdg_chain_set", "gdg_process", "gdg_process_subset", "gdg_process_device",
    "gdg_staging_buffers", "gdg_process_staged", "gdg_device_alloc", "gdg_device_free", "gdg_copy_to_device", "gdg_copy_to_host", "gdg_copy_rows_device", "gdg_fft_real", "gdg_fft_real_inverse", "gdg_debug_oversample_decimate", "gdg_profile_enable",
    "gdg_profile_read", "gdg_tuner_enqueue", "gdg_tuner_enqueue_device", "gdg_tuner_enqueue_staged", "gdg_tuner_analyze", "gdg_tuner_note_name",
    "gdg_spatializer_set_position", "gdg_spatializer_set_sample_rate", "gdg_spatialize", "gdg_spatialize_device", "gdg_spatialize_staged",
    "gdg_wave_bytes_per_sample", "gdg_wave_decode", "gdg_wave_decode_device", "gdg_wave_encode", "gdg_wave_encode_device",
    "gdg_resample_time_length", "gdg_resample_time", "gdg_resample_time_device",
    "gdg_meter_configure", "gdg_meter_set_enabled", "gdg_meter_process", "gdg_meter_process_device", "gdg_meter_analyze", "gdg_meter_state",
    "gdg_metronome_set_tick", "gdg_metronome_set_tock", "gdg_metronome_configure", "gdg_metronome_process", "gdg_metronome_process_device",
    "gdg_batch_length", "gdg_batch_run", "gdg_batch_run_shard", "gdg_batch_finish_master", "gdg_batch_release", "gdg_profile_sample", "gdg_ctx_set_window", "gdg_process_window_device", "gdg_ctx_set_overlap",
    "gdg_ctx_set_option", "gdg_ctx_get_option", "gdg_option_count", "gdg_option_name", "gdg_numa_probe", "gdg_ctx_trim", "gdg_tuner_replace",
]


def option_names():
    """The keys gdg_ctx_set_option understands."""
    return [lib().gdg_option_name(i).decode() for i in range(lib().gdg_option_count())]


def numa_probe(sysfs_root, pci_bus_id, capacity=4096):
    """(node, [cpus]) of a PCI device from sysfs (gdg_numa_probe; no device needed)."""
    node, n = C.c_int(-1), C.c_int(0)
    cpus = (C.c_int * capacity)()
    rc = lib().gdg_numa_probe(sysfs_root.encode(), pci_bus_id.encode(), C.byref(node), cpus, capacity, C.byref(n))
    if rc != GDG_OK:
        raise GdgError(rc, "gdg_numa_probe")
    return int(node.value), [int(cpus[i]) for i in range(min(capacity, n.value))]


class GdgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gdg error %d: %s" % (code, msg))
        self.code = code


class TunerResult(C.Structure):
    _fields_ = [("frequency", C.c_double), ("note_index", C.c_int32), ("cents", C.c_int8)]


class BatchInput(C.Structure):          # gdg_batch_input
    _fields_ = [("bytes", C.c_void_p), ("samples_per_channel", C.c_size_t), ("format", C.c_int), ("sample_rate", C.c_uint32),
                ("channels", C.c_uint), ("channel", C.c_uint)]


class BatchShardOut(C.Structure):       # gdg_batch_shard_out
    _fields_ = [("master_left", C.c_void_p), ("master_right", C.c_void_p), ("metronome_bytes", C.c_void_p), ("metronome", C.c_void_p),
                ("job_samples", C.c_size_t)]


class BatchOptions(C.Structure):        # gdg_batch_options
    _fields_ = [("target_rate", C.c_uint32), ("out_format", C.c_int), ("metronome_to_master", C.c_int), ("run_meters", C.c_int),
                ("tuner_enqueue", C.c_int)]


def build(force=False):
    """Compile libgdg.so for gfx950 with hipcc (recipe: csrc/Makefile).  Works without a GPU."""
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-C", CSRC])
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "host")])     # C++ mirror of effects.Unit / signal.Chain
    return LIB_PATH


_lib = None


def lib():
    """The loaded libgdg.so (raises if it was never built: there is no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GdgError(GDG_ERR_NO_DEVICE, "libgdg.so is not built (run __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp, i32, u32, dbl = C.c_void_p, C.c_int, C.c_uint32, C.c_double
        sig = {
            "gdg_version": (C.c_char_p, []),
            "gdg_device_count": (i32, []),
            "gdg_ctx_create": (i32, [i32, i32, i32, C.POINTER(vp)]),
            "gdg_ctx_destroy": (i32, [vp]),
            "gdg_last_error": (C.c_char_p, [vp]),
            "gdg_ctx_channels": (i32, [vp]),
            "gdg_ctx_share_ir_spectra": (i32, [vp, i32]),
            "gdg_ctx_stream": (vp, [vp]),
            "gdg_ctx_synchronize": (i32, [vp]),
            "gdg_unit_create": (i32, [vp, i32, i32, C.POINTER(i32)]),
            "gdg_unit_destroy": (i32, [vp, i32]),
            "gdg_unit_set_param": (i32, [vp, i32, i32, C.c_int32]),
            "gdg_unit_get_param": (i32, [vp, i32, i32, C.POINTER(C.c_int32)]),
            "gdg_unit_set_fir": (i32, [vp, i32, vp, i32]),
            "gdg_unit_reset": (i32, [vp, i32]),
            "gdg_unit_compile_fir": (i32, [vp, i32, i32, vp, vp, vp, vp, u32]),
            "gdg_unit_get_fir": (i32, [vp, i32, vp, i32, C.POINTER(i32)]),
            "gdg_chain_set": (i32, [vp, i32, vp, vp, i32]),
            "gdg_process": (i32, [vp, vp, vp, i32, u32]),
            "gdg_process_subset": (i32, [vp, vp, i32, vp, vp, i32, u32]),
            "gdg_process_device": (i32, [vp, vp, vp, i32, u32]),
            "gdg_staging_buffers": (i32, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32)]),
            "gdg_process_staged": (i32, [vp, vp, i32, i32, u32]),
            "gdg_device_alloc": (i32, [vp, C.c_size_t, C.POINTER(vp)]),
            "gdg_device_free": (i32, [vp, vp]),
            "gdg_copy_to_device": (i32, [vp, vp, vp, C.c_size_t]),
            "gdg_copy_to_host": (i32, [vp, vp, vp, C.c_size_t]),
            "gdg_copy_rows_device": (i32, [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t]),
            "gdg_fft_real": (i32, [vp, vp, i32, vp]),
            "gdg_fft_real_inverse": (i32, [vp, vp, i32, vp]),
            "gdg_debug_oversample_decimate": (i32, [vp, i32, vp, i32, vp, vp, vp]),
            "gdg_profile_enable": (i32, [vp, i32]),
            "gdg_profile_read": (i32, [vp, i32, C.POINTER(dbl), C.POINTER(i32)]),
            "gdg_tuner_enqueue": (i32, [vp, vp, i32, u32]),
            "gdg_tuner_enqueue_device": (i32, [vp, vp, i32, u32]),
            "gdg_tuner_enqueue_staged": (i32, [vp, i32, u32]),
            "gdg_tuner_analyze": (i32, [vp, C.POINTER(TunerResult)]),
            "gdg_tuner_note_name": (C.c_char_p, [i32]),
            "gdg_spatializer_set_position": (i32, [vp, i32, dbl, dbl, dbl]),
            "gdg_spatializer_set_sample_rate": (i32, [vp, u32]),
            "gdg_spatialize": (i32, [vp, vp, vp, vp, i32]),
            "gdg_spatialize_device": (i32, [vp, vp, vp, i32]),
            "gdg_spatialize_staged": (i32, [vp, i32, vp, vp, i32]),
            "gdg_wave_bytes_per_sample": (i32, [i32]),
            "gdg_wave_decode": (i32, [vp, i32, vp, C.c_size_t, C.c_uint, vp]),
            "gdg_wave_decode_device": (i32, [vp, i32, vp, C.c_size_t, C.c_uint, vp]),
            "gdg_wave_encode": (i32, [vp, i32, vp, C.c_size_t, C.c_uint, vp]),
            "gdg_wave_encode_device": (i32, [vp, i32, vp, C.c_size_t, C.c_uint, vp]),
            "gdg_resample_time_length": (i32, [i32, u32, u32]),
            "gdg_resample_time": (i32, [vp, vp, i32, u32, u32, vp, i32]),
            "gdg_resample_time_device": (i32, [vp, vp, i32, u32, u32, vp, i32]),
            "gdg_meter_configure": (i32, [vp, i32]),
            "gdg_meter_set_enabled": (i32, [vp, i32, i32]),
            "gdg_meter_process": (i32, [vp, vp, i32, u32]),
            "gdg_meter_process_device": (i32, [vp, vp, C.c_size_t, i32, u32]),
            "gdg_meter_analyze": (i32, [vp, vp, vp]),
            "gdg_meter_state": (i32, [vp, i32, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(C.c_uint64)]),
            "gdg_metronome_set_tick": (i32, [vp, vp, i32]),
            "gdg_metronome_set_tock": (i32, [vp, vp, i32]),
            "gdg_metronome_configure": (i32, [vp, u32, u32, u32]),
            "gdg_metronome_process": (i32, [vp, vp, i32]),
            "gdg_metronome_process_device": (i32, [vp, vp, i32]),
            "gdg_ctx_set_window": (i32, [vp, i32]),
            "gdg_ctx_set_overlap": (i32, [vp, i32]),
            "gdg_process_window_device": (i32, [vp, vp, vp, C.c_size_t, i32, u32]),
            "gdg_batch_length": (i32, [vp, vp, i32, u32, C.POINTER(C.c_size_t)]),
            "gdg_batch_run": (i32, [vp, vp, i32, vp, vp]),
            "gdg_batch_release": (i32, [vp]),
            "gdg_batch_run_shard": (i32, [vp, vp, i32, vp, vp, vp]),
            "gdg_batch_finish_master": (i32, [vp, i32, vp, vp, i32, vp, C.c_size_t, u32, i32, vp, vp]),
            "gdg_profile_sample": (i32, [vp, i32]),
            "gdg_ctx_set_option": (i32, [vp, C.c_char_p, C.c_longlong]),
            "gdg_ctx_get_option": (i32, [vp, C.c_char_p, C.POINTER(C.c_longlong)]),
            "gdg_ctx_trim": (i32, [vp]),
            "gdg_tuner_replace": (i32, [vp, i32, vp, i32, u32]),
            "gdg_option_count": (i32, []),
            "gdg_option_name": (C.c_char_p, [i32]),
            "gdg_numa_probe": (i32, [C.c_char_p, C.c_char_p, C.POINTER(i32), C.POINTER(i32), i32, C.POINTER(i32)]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def device_count():
    return lib().gdg_device_count()


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class DeviceBuffer:
    """A [rows][cols] float64 array in the context's device memory (plain device pointer underneath)."""

    def __init__(self, ctx, rows, cols):
        self.ctx, self.rows, self.cols = ctx, rows, cols
        p = C.c_void_p()
        ctx._check(lib().gdg_device_alloc(ctx._h, rows * cols * 8, C.byref(p)))
        self.ptr = p.value

    def upload(self, a):
        a = _f64(a)
        assert a.size == self.rows * self.cols
        self.ctx._check(lib().gdg_copy_to_device(self.ctx._h, self.ptr, a.ctypes.data, a.nbytes))

    def download(self):
        out = np.empty((self.rows, self.cols), dtype=np.float64)
        self.ctx._check(lib().gdg_copy_to_host(self.ctx._h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().gdg_device_free(self.ctx._h, self.ptr)
            self.ptr = None


class Context:
    """One shard of channels on one GPU (gdg_ctx)."""

    def __init__(self, n_channels, max_frames=8192, device=0):
        self.n_channels, self.max_frames, self.device = n_channels, max_frames, device
        h = C.c_void_p()
        rc = lib().gdg_ctx_create(n_channels, max_frames, device, C.byref(h))
        if rc != GDG_OK:
            raise GdgError(rc, "gdg_ctx_create failed (no usable HIP device?)")
        self._h = h
        self._chains = [[] for _ in range(n_channels)]      # [(handle, bypass)]

    def close(self):
        if getattr(self, "_h", None):
            lib().gdg_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def share_ir_spectra(self, enable):
        self._check(lib().gdg_ctx_share_ir_spectra(self._h, 1 if enable else 0))

    def _check(self, rc):
        if rc != GDG_OK:
            raise GdgError(rc, lib().gdg_last_error(self._h).decode())

    # -- units / chains ------------------------------------------------------------------------
    def unit_create(self, channel, unit_type):
        if isinstance(unit_type, str):
            unit_type = UNIT[unit_type]
        h = C.c_int(-1)
        self._check(lib().gdg_unit_create(self._h, channel, unit_type, C.byref(h)))
        return h.value

    def unit_destroy(self, handle):
        self._check(lib().gdg_unit_destroy(self._h, handle))

    def unit_set_param(self, handle, idx, value):
        self._check(lib().gdg_unit_set_param(self._h, handle, idx, int(value)))

    def unit_get_param(self, handle, idx):
        v = C.c_int32(0)
        self._check(lib().gdg_unit_get_param(self._h, handle, idx, C.byref(v)))
        return v.value

    def unit_set_fir(self, handle, taps):
        t = _f64(taps)
        self._check(lib().gdg_unit_set_fir(self._h, handle, t.ctypes.data if t.size else None, t.size))

    def unit_compile_fir(self, handle, filters, target_order=0):
        """filters: list of (taps or None, gain_compensation_factor, level_db) per slot (gdg_unit_compile_fir)."""
        n = len(filters)
        arrs = [(_f64(t) if t is not None else None) for t, _, _ in filters]
        ptrs = (C.c_void_p * n)(*[(a.ctypes.data if a is not None and a.size else None) for a in arrs])
        lens = (C.c_int * n)(*[(a.size if a is not None else 0) for a in arrs])
        comp = (C.c_double * n)(*[float(f[1]) for f in filters])
        lev = (C.c_int32 * n)(*[int(f[2]) for f in filters])
        self._check(lib().gdg_unit_compile_fir(self._h, handle, n, ptrs, lens, comp, lev, target_order))

    def unit_get_fir(self, handle):
        n = C.c_int(0)
        self._check(lib().gdg_unit_get_fir(self._h, handle, None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=np.float64)
        if n.value:
            self._check(lib().gdg_unit_get_fir(self._h, handle, out.ctypes.data, n.value, C.byref(n)))
        return out

    def unit_reset(self, handle):
        self._check(lib().gdg_unit_reset(self._h, handle))

    def chain_set(self, channel, handles, bypass=None):
        n = len(handles)
        bypass = [False] * n if bypass is None else bypass
        hs = (C.c_int * max(n, 1))(*handles)
        bs = (C.c_uint8 * max(n, 1))(*[1 if b else 0 for b in bypass])
        self._check(lib().gdg_chain_set(self._h, channel, hs, bs, n))
        self._chains[channel] = list(zip(handles, bypass))

    def append_unit(self, channel, unit_type, params=None, fir=None, bypass=False):
        """AppendUnit + SetBypass + parameter set-up in one go (test convenience)."""
        h = self.unit_create(channel, unit_type)
        if params is not None:
            for i, v in enumerate(params):
                self.unit_set_param(h, i, v)
        if fir is not None:
            self.unit_set_fir(h, fir)
        chain = self._chains[channel] + [(h, bypass)]
        self.chain_set(channel, [c[0] for c in chain], [c[1] for c in chain])
        return h

    # -- processing --------------------------------------------------------------------------------
    def process(self, x, sample_rate):
        """x: [n_channels][frames] host array -> same-shaped output (gdg_process, blocking)."""
        x = _f64(x)
        assert x.ndim == 2 and x.shape[0] == self.n_channels
        frames = x.shape[1]
        out = np.empty_like(x)
        ins = (C.c_void_p * self.n_channels)(*[x[c].ctypes.data for c in range(self.n_channels)])
        outs = (C.c_void_p * self.n_channels)(*[out[c].ctypes.data for c in range(self.n_channels)])
        self._check(lib().gdg_process(self._h, ins, outs, frames, sample_rate))
        return out

    def process_subset(self, channels, x, sample_rate):
        """x: [len(channels)][frames]; only the listed channels' chains run (gdg_process_subset)."""
        x = _f64(x)
        n = len(channels)
        assert x.ndim == 2 and x.shape[0] == n
        out = np.empty_like(x)
        chans = (C.c_int * n)(*channels)
        ins = (C.c_void_p * n)(*[x[i].ctypes.data for i in range(n)])
        outs = (C.c_void_p * n)(*[out[i].ctypes.data for i in range(n)])
        self._check(lib().gdg_process_subset(self._h, chans, n, ins, outs, x.shape[1], sample_rate))
        return out

    def process_staged(self, channels, x, sample_rate):
        """The cgo-friendly path: copy frames into the pinned slab rows, gdg_process_staged, copy rows out."""
        x = _f64(x)
        n, frames = x.shape
        pin, pout, stride = C.c_void_p(), C.c_void_p(), C.c_int(0)
        self._check(lib().gdg_staging_buffers(self._h, C.byref(pin), C.byref(pout), C.byref(stride)))
        slab_in = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_double)), shape=(self.n_channels, stride.value))
        slab_out = np.ctypeslib.as_array(C.cast(pout, C.POINTER(C.c_double)), shape=(self.n_channels, stride.value))
        idx = np.asarray(channels, dtype=np.intp)
        slab_in[idx, :frames] = x
        chans = (C.c_int * n)(*channels)
        self._check(lib().gdg_process_staged(self._h, chans, n, frames, sample_rate))
        return slab_out[idx, :frames].copy()

    def process_device(self, d_in, d_out, frames, sample_rate):
        """Device-resident block; d_in / d_out are plain device pointers (ints) or DeviceBuffers."""
        pi = d_in.ptr if isinstance(d_in, DeviceBuffer) else d_in
        po = d_out.ptr if isinstance(d_out, DeviceBuffer) else d_out
        self._check(lib().gdg_process_device(self._h, pi, po, frames, sample_rate))

    def synchronize(self):
        self._check(lib().gdg_ctx_synchronize(self._h))

    def alloc(self, rows, cols):
        return DeviceBuffer(self, rows, cols)

    @property
    def stream(self):
        return lib().gdg_ctx_stream(self._h)

    # -- profiling ------------------------------------------------------------------------------------
    def profile_enable(self, on=True, kinds=None):
        """on: every kernel launch; kinds: only the listed kernel kinds (cheaper inside a timed region)."""
        mask = sum(1 << (k + 1) for k in kinds) if kinds else (1 if on else 0)
        self._check(lib().gdg_profile_enable(self._h, mask))

    def profile_sample(self, every):
        self._check(lib().gdg_profile_sample(self._h, every))

    def profile_read(self, kind):
        ms, n = C.c_double(0.0), C.c_int(0)
        self._check(lib().gdg_profile_read(self._h, kind, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # -- the transforms underneath the power amp -------------------------------------------------------
    def fft_real(self, x):
        """fft.RealFourier: n reals -> n / 2 + 1 complex bins."""
        x = _f64(x)
        out = np.empty(2 * (x.size // 2 + 1), dtype=np.float64)
        self._check(lib().gdg_fft_real(self._h, x.ctypes.data, x.size, out.ctypes.data))
        return out.view(np.complex128)

    def debug_oversample_decimate(self, factor, x, state):
        """OversamplerDecimator.Oversample then Decimate on the HIP tiles (debug entry); `state` (8 + taps - 1 doubles, zeros = fresh) is updated in place.
        Returns (oversampled, decimated)."""
        x = _f64(x)
        assert state.dtype == np.float64 and state.flags.c_contiguous
        up, down = np.empty(factor * x.size), np.empty(x.size)
        self._check(lib().gdg_debug_oversample_decimate(self._h, factor, x.ctypes.data, x.size, state.ctypes.data, up.ctypes.data, down.ctypes.data))
        return up, down

    def fft_real_inverse(self, spectrum, n):
        s = np.ascontiguousarray(spectrum, dtype=np.complex128)
        assert s.size == n // 2 + 1
        out = np.empty(n, dtype=np.float64)
        self._check(lib().gdg_fft_real_inverse(self._h, s.ctypes.data, n, out.ctypes.data))
        return out

    # -- tuner / spatializer ---------------------------------------------------------------------------
    def tuner_enqueue(self, x, sample_rate):
        x = _f64(x)
        assert x.ndim == 2 and x.shape[0] == self.n_channels
        ptrs = (C.c_void_p * self.n_channels)(*[x[c].ctypes.data for c in range(self.n_channels)])
        self._check(lib().gdg_tuner_enqueue(self._h, ptrs, x.shape[1], sample_rate))

    def tuner_enqueue_device(self, d_x, frames, sample_rate):
        p = d_x.ptr if isinstance(d_x, DeviceBuffer) else d_x
        self._check(lib().gdg_tuner_enqueue_device(self._h, p, frames, sample_rate))

    def tuner_analyze(self, raw=False):
        """raw=True: the C structs as the call left them (what a C or Go caller gets; building 256 dicts costs Python ~0.1 ms)"""
        res = (TunerResult * self.n_channels)()
        self._check(lib().gdg_tuner_analyze(self._h, res))
        if raw:
            return res
        return [{"frequency": r.frequency, "note_index": r.note_index, "cents": r.cents,
                 "note": lib().gdg_tuner_note_name(r.note_index).decode()} for r in res]

    def spatializer_set_position(self, channel, azimuth, distance, level):
        self._check(lib().gdg_spatializer_set_position(self._h, channel, azimuth, distance, level))

    def spatializer_set_sample_rate(self, rate):
        self._check(lib().gdg_spatializer_set_sample_rate(self._h, rate))

    def spatialize(self, x):
        x = _f64(x)
        assert x.ndim == 2 and x.shape[0] == self.n_channels
        n = x.shape[1]
        ptrs = (C.c_void_p * self.n_channels)(*[x[c].ctypes.data for c in range(self.n_channels)])
        left, right = np.empty(n), np.empty(n)
        self._check(lib().gdg_spatialize(self._h, ptrs, left.ctypes.data, right.ctypes.data, n))
        return left, right

    def spatialize_device(self, d_x, d_out_lr, frames):
        pi = d_x.ptr if isinstance(d_x, DeviceBuffer) else d_x
        po = d_out_lr.ptr if isinstance(d_out_lr, DeviceBuffer) else d_out_lr
        self._check(lib().gdg_spatialize_device(self._h, pi, po, frames))

    # -- data formats either side of the path -------------------------------------------
    def wave_decode(self, fmt, data, channels=1):
        """Data section of a WAVE file (interleaved bytes) -> planar float64 [channels][n]."""
        f = WAVE_FORMATS[fmt] if isinstance(fmt, str) else fmt
        data = np.ascontiguousarray(data, dtype=np.uint8)
        w = lib().gdg_wave_bytes_per_sample(f)
        per = data.size // (w * channels) if w else 0
        out = np.empty((channels, per), dtype=np.float64)
        self._check(lib().gdg_wave_decode(self._h, f, data.ctypes.data, per, channels, out.ctypes.data))
        return out[0] if channels == 1 else out

    def wave_encode(self, fmt, samples):
        """Planar float64 [channels][n] (or [n]) -> interleaved little-endian bytes."""
        f = WAVE_FORMATS[fmt] if isinstance(fmt, str) else fmt
        x = _f64(samples)
        channels, per = (1, x.size) if x.ndim == 1 else x.shape
        out = np.empty(channels * per * max(lib().gdg_wave_bytes_per_sample(f), 1), dtype=np.uint8)
        self._check(lib().gdg_wave_encode(self._h, f, x.ctypes.data, per, channels, out.ctypes.data))
        return out

    def resample_time(self, samples, source_rate, target_rate):
        x = _f64(samples)
        n_out = lib().gdg_resample_time_length(x.size, source_rate, target_rate)
        out = np.empty(max(n_out, 0), dtype=np.float64)
        self._check(lib().gdg_resample_time(self._h, x.ctypes.data, x.size, source_rate, target_rate, out.ctypes.data, n_out))
        return out

    def meter_configure(self, n_ports):
        self._n_ports = n_ports
        self._check(lib().gdg_meter_configure(self._h, n_ports))

    def meter_set_enabled(self, enabled, port=-1):
        self._check(lib().gdg_meter_set_enabled(self._h, port, 1 if enabled else 0))

    def meter_process(self, x, sample_rate):
        x = _f64(x)
        assert x.ndim == 2 and x.shape[0] == self._n_ports
        ptrs = (C.c_void_p * self._n_ports)(*[x[p].ctypes.data for p in range(self._n_ports)])
        self._check(lib().gdg_meter_process(self._h, ptrs, x.shape[1], sample_rate))

    def meter_process_device(self, d_rows, row_stride, frames, sample_rate):
        p = d_rows.ptr if isinstance(d_rows, DeviceBuffer) else d_rows
        self._check(lib().gdg_meter_process_device(self._h, p, row_stride, frames, sample_rate))

    def meter_analyze(self):
        lv = np.empty(self._n_ports, dtype=np.int32)
        pk = np.empty(self._n_ports, dtype=np.int32)
        self._check(lib().gdg_meter_analyze(self._h, lv.ctypes.data, pk.ctypes.data))
        return lv, pk

    def meter_state(self, port):
        c, p, n = C.c_double(), C.c_double(), C.c_uint64()
        self._check(lib().gdg_meter_state(self._h, port, C.byref(c), C.byref(p), C.byref(n)))
        return c.value, p.value, n.value

    def metronome_set_sounds(self, tick, tock):
        for fn, a in ((lib().gdg_metronome_set_tick, tick), (lib().gdg_metronome_set_tock, tock)):
            if a is None:
                self._check(fn(self._h, None, 0))
            else:
                a = _f64(a)
                self._check(fn(self._h, a.ctypes.data if a.size else C.cast(C.create_string_buffer(8), C.c_void_p), a.size))

    def metronome_configure(self, beats_per_period, bpm_speed, sample_rate):
        self._check(lib().gdg_metronome_configure(self._h, beats_per_period, bpm_speed, sample_rate))

    def set_window(self, frames_per_call):
        """Time blocking: up to `frames_per_call` (1, 2, 4, 8, 16) consecutive 8192-sample frames per channel and call."""
        self._check(lib().gdg_ctx_set_window(self._h, frames_per_call))

    def tuner_replace(self, channel, samples, sample_rate):
        """One channel's whole 96000-sample ring, oldest first (gdg_tuner_replace)."""
        a = np.ascontiguousarray(samples, dtype=np.float64)
        self._check(lib().gdg_tuner_replace(self._h, channel, a.ctypes.data, a.size, sample_rate))

    def trim(self):
        """Give spare device memory back (gdg_ctx_trim); blocks."""
        self._check(lib().gdg_ctx_trim(self._h))

    def set_option(self, key, value):
        """Launch-shape options (include/gdg.h, gdg_ctx_set_option): what used to be environment variables."""
        self._check(lib().gdg_ctx_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_longlong(0)
        self._check(lib().gdg_ctx_get_option(self._h, key.encode(), C.byref(v)))
        return int(v.value)

    def set_overlap(self, groups):
        """Channel groups of the device-resident calls, free-running on streams of their own (include/gdg.h)."""
        self._check(lib().gdg_ctx_set_overlap(self._h, groups))

    def process_window_device(self, d_in, d_out, row_stride, frames_in_window, sample_rate):
        pi = d_in.ptr if isinstance(d_in, DeviceBuffer) else d_in
        po = d_out.ptr if isinstance(d_out, DeviceBuffer) else d_out
        self._check(lib().gdg_process_window_device(self._h, pi, po, row_stride, frames_in_window, sample_rate))

    def _batch_inputs(self, inputs):
        n = len(inputs)
        arr = (BatchInput * n)()
        keep = []
        for i, it in enumerate(inputs):
            if it is None:
                continue
            data, fmt, rate = it[0], it[1], it[2]
            channels, channel = (it[3], it[4]) if len(it) > 3 else (1, 0)
            f = WAVE_FORMATS[fmt] if isinstance(fmt, str) else fmt
            data = np.ascontiguousarray(data, dtype=np.uint8)
            keep.append(data)
            w = max(lib().gdg_wave_bytes_per_sample(f), 1)
            arr[i] = BatchInput(data.ctypes.data if data.size else None, data.size // (w * max(channels, 1)), f, rate, channels, channel)
        return arr, keep

    def batch_length(self, inputs, target_rate):
        arr, _keep = self._batch_inputs(inputs)
        length = C.c_size_t(0)
        self._check(lib().gdg_batch_length(self._h, arr, len(inputs), target_rate, C.byref(length)))
        return length.value

    def batch_run(self, inputs, target_rate, out_format, metronome_to_master=False, run_meters=False, tuner_enqueue=False, outs=None):
        """controller.processFiles on the device (controller/controller.go:2809-3219 without prompts and file I/O).
        inputs: per channel None ("leaving channel empty") or (data-section bytes, format, sample_rate[, channels, channel]);
        returns the N + 3 output data sections (uint8 arrays): out_0 .. out_{N-1}, master left, master right, metronome."""
        n = len(inputs)
        arr, _keep = self._batch_inputs(inputs)
        fo = WAVE_FORMATS[out_format] if isinstance(out_format, str) else out_format
        opt = BatchOptions(target_rate, fo, int(bool(metronome_to_master)), int(bool(run_meters)), int(bool(tuner_enqueue)))
        length = C.c_size_t(0)
        self._check(lib().gdg_batch_length(self._h, arr, n, target_rate, C.byref(length)))
        wo = lib().gdg_wave_bytes_per_sample(fo)
        if outs is None:
            outs = [np.zeros(length.value * wo, dtype=np.uint8) for _ in range(n + 3)]
        assert len(outs) == n + 3 and all(o.dtype == np.uint8 and o.size == length.value * wo for o in outs)
        ptrs = (C.c_void_p * (n + 3))(*[(o.ctypes.data if o.size else None) for o in outs])
        self._check(lib().gdg_batch_run(self._h, arr, n, C.byref(opt), ptrs))
        return outs

    def batch_prepared(self, inputs, target_rate, out_format, metronome_to_master=False, run_meters=False, tuner_enqueue=False):
        """gdg_batch_run with its arguments marshalled ONCE: returns (call, outs); call() is the C call alone (what a C or Go caller pays --
        filling 512 input structs and 515 pointers in Python costs ~3 ms per run), outs the N + 3 output buffers it writes."""
        n = len(inputs)
        arr, keep = self._batch_inputs(inputs)
        fo = WAVE_FORMATS[out_format] if isinstance(out_format, str) else out_format
        opt = BatchOptions(target_rate, fo, int(bool(metronome_to_master)), int(bool(run_meters)), int(bool(tuner_enqueue)))
        length = C.c_size_t(0)
        self._check(lib().gdg_batch_length(self._h, arr, n, target_rate, C.byref(length)))
        wo = lib().gdg_wave_bytes_per_sample(fo)
        outs = [np.zeros(length.value * wo, dtype=np.uint8) for _ in range(n + 3)]
        ptrs = (C.c_void_p * (n + 3))(*[(o.ctypes.data if o.size else None) for o in outs])
        fn, h, ref = lib().gdg_batch_run, self._h, C.byref(opt)

        def call(_keep=(keep, arr, opt, ptrs, outs)):
            self._check(fn(h, arr, n, ref, ptrs))
        return call, outs

    def batch_shard_prepared(self, inputs, target_rate, out_format, job_samples=0, metronome=False, run_meters=False, tuner_enqueue=False):
        """gdg_batch_run_shard with its arguments marshalled and its result buffers allocated ONCE: returns (call, result) where call() is the C
        call alone and result = (outs, left, right, metronome_bytes, metronome_f64) as batch_run_shard returns them."""
        n = len(inputs)
        arr, keep = self._batch_inputs(inputs)
        fo = WAVE_FORMATS[out_format] if isinstance(out_format, str) else out_format
        opt = BatchOptions(target_rate, fo, 0, int(bool(run_meters)), int(bool(tuner_enqueue)))
        length = job_samples or self.batch_length(inputs, target_rate)
        wo = lib().gdg_wave_bytes_per_sample(fo)
        outs = [np.zeros(length * wo, dtype=np.uint8) for _ in range(n)]
        left, right = np.zeros(max(length, 1)), np.zeros(max(length, 1))
        mb = np.zeros(length * wo, dtype=np.uint8) if metronome else None
        mf = np.zeros(length) if metronome else None
        ptrs = (C.c_void_p * n)(*[(o.ctypes.data if o.size else None) for o in outs])
        so = BatchShardOut(left.ctypes.data, right.ctypes.data, mb.ctypes.data if (metronome and length) else None,
                           mf.ctypes.data if (metronome and length) else None, job_samples)
        fn, h, ropt, rso = lib().gdg_batch_run_shard, self._h, C.byref(opt), C.byref(so)

        def call(_keep=(keep, arr, opt, ptrs, so, outs, left, right, mb, mf)):
            self._check(fn(h, arr, n, ropt, ptrs, rso))
        return call, (outs, left[:length], right[:length], mb, mf)

    def batch_run_shard(self, inputs, target_rate, out_format, job_samples=0, metronome=False, run_meters=False, tuner_enqueue=False, outs=None):
        """One shard of a batch split over several contexts (gdg_batch_run_shard): returns (outs, left, right, metronome_bytes,
        metronome_f64): the shard's n encoded chain outputs, its float64 partial master mix, and -- on the shard that runs the
        metronome -- the encoded metronome track and its float64 samples (the master's aux input)."""
        n = len(inputs)
        arr, _keep = self._batch_inputs(inputs)
        fo = WAVE_FORMATS[out_format] if isinstance(out_format, str) else out_format
        opt = BatchOptions(target_rate, fo, 0, int(bool(run_meters)), int(bool(tuner_enqueue)))
        length = job_samples or self.batch_length(inputs, target_rate)
        wo = lib().gdg_wave_bytes_per_sample(fo)
        if outs is None:
            outs = [np.zeros(length * wo, dtype=np.uint8) for _ in range(n)]
        left, right = np.zeros(length), np.zeros(length)
        mb = np.zeros(length * wo, dtype=np.uint8) if metronome else None
        mf = np.zeros(length) if metronome else None
        ptrs = (C.c_void_p * n)(*[(o.ctypes.data if o.size else None) for o in outs])
        so = BatchShardOut(left.ctypes.data if length else None, right.ctypes.data if length else None,
                           mb.ctypes.data if (metronome and length) else None, mf.ctypes.data if (metronome and length) else None, job_samples)
        if length == 0:
            so = BatchShardOut(left.ctypes.data_as(C.c_void_p), right.ctypes.data_as(C.c_void_p), None, None, job_samples)
        self._check(lib().gdg_batch_run_shard(self._h, arr, n, C.byref(opt), ptrs, C.byref(so)))
        return outs, left, right, mb, mf

    def batch_finish_master(self, out_format, lefts, rights, aux=None, sample_rate=0, run_meters=False):
        """master = sum of the shards' partial mixes (in shard order) + aux, encoded on this context's device."""
        G, n = len(lefts), lefts[0].size
        fo = WAVE_FORMATS[out_format] if isinstance(out_format, str) else out_format
        wo = lib().gdg_wave_bytes_per_sample(fo)
        lefts = [_f64(a) for a in lefts]
        rights = [_f64(a) for a in rights]
        lp = (C.c_void_p * G)(*[a.ctypes.data for a in lefts])
        rp = (C.c_void_p * G)(*[a.ctypes.data for a in rights])
        ml, mr = np.zeros(n * wo, dtype=np.uint8), np.zeros(n * wo, dtype=np.uint8)
        a = _f64(aux) if aux is not None else None
        self._check(lib().gdg_batch_finish_master(self._h, fo, lp, rp, G, a.ctypes.data if a is not None else None, n, sample_rate,
                                                  int(bool(run_meters)), ml.ctypes.data if n else None, mr.ctypes.data if n else None))
        return ml, mr

    def batch_release(self):
        self._check(lib().gdg_batch_release(self._h))

    def metronome_process(self, frames):
        out = np.empty(frames, dtype=np.float64)
        self._check(lib().gdg_metronome_process(self._h, out.ctypes.data, frames))
        return out
