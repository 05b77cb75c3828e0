"""ctypes binding of libgdg_host.so: the C++ mirror of effects.Unit / signal.Chain (host/gdg_host.hpp).

Test and bench plumbing only; the mirrored interface itself is the C++ one (and its Go twin in go/).
A Go `error` comes back as a Python exception (HostError) carrying the reference's message text.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgdg_host.so")


class HostError(RuntimeError):
    pass


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "host")])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HostError("libgdg_host.so is not built")
        # libgdg.so first so that the host library's NEEDED entry resolves to the in-tree copy
        C.CDLL(os.path.join(_HERE, "lib", "libgdg.so"), mode=C.RTLD_GLOBAL)
        L = C.CDLL(LIB_PATH)
        vp, i32, cs = C.c_void_p, C.c_int, C.c_char_p
        sig = {
            "gdgh_engine_create": (vp, [i32, i32, i32]), "gdgh_engine_destroy": (None, [vp]),
            "gdgh_engine_set_rendezvous": (None, [vp, i32, i32]), "gdgh_engine_last_error": (cs, [vp]),
            "gdgh_engine_process_all": (cs, [vp, vp, vp, i32, C.c_uint32]),
            "gdgh_engine_batch_run": (cs, [vp, vp, i32, vp, i32, vp, C.POINTER(C.c_size_t)]),
            "gdgh_engine_context": (vp, [vp, i32]), "gdgh_engine_shard_range": (None, [vp, i32, C.POINTER(i32), C.POINTER(i32)]),
            "gdgh_engine_create_sharded": (vp, [i32, i32, vp, i32]), "gdgh_engine_shards": (i32, [vp]), "gdgh_engine_shard_of": (i32, [vp, i32]),
            "gdgh_spatializer_create": (vp, [vp, C.c_uint32]), "gdgh_spatializer_destroy": (None, [vp]),
            "gdgh_spatializer_set": (cs, [vp, i32, C.c_uint32, C.c_double]), "gdgh_spatializer_get": (cs, [vp, i32, C.c_uint32, C.POINTER(C.c_double)]),
            "gdgh_spatializer_input_count": (C.c_uint32, [vp]), "gdgh_spatializer_output_count": (C.c_uint32, [vp]),
            "gdgh_spatializer_set_sample_rate": (None, [vp, C.c_uint32]),
            "gdgh_spatializer_process": (None, [vp, vp, vp, vp, vp, i32, i32]),
            "gdgh_tuner_create": (vp, [i32]), "gdgh_tuner_destroy": (None, [vp]), "gdgh_tuner_process": (None, [vp, vp, i32, C.c_uint32]),
            "gdgh_tuner_analyze": (cs, [vp, C.POINTER(i32), C.POINTER(C.c_double), vp, i32]),
            "gdgh_irs_create": (vp, []), "gdgh_irs_destroy": (None, [vp]),
            "gdgh_irs_add": (None, [vp, cs, C.c_uint32, C.c_int32, vp, i32]),
            "gdgh_chain_create": (vp, [vp, vp]), "gdgh_chain_destroy": (None, [vp]),
            "gdgh_chain_append_unit": (cs, [vp, i32, C.POINTER(i32)]), "gdgh_chain_remove_unit": (cs, [vp, i32]),
            "gdgh_chain_move_up": (cs, [vp, i32]), "gdgh_chain_move_down": (cs, [vp, i32]),
            "gdgh_chain_unit_type": (cs, [vp, i32, C.POINTER(i32)]), "gdgh_chain_set_bypass": (cs, [vp, i32, i32]),
            "gdgh_chain_get_bypass": (cs, [vp, i32, C.POINTER(i32)]),
            "gdgh_chain_set_discrete": (cs, [vp, i32, cs, cs]), "gdgh_chain_get_discrete": (cs, [vp, i32, cs, vp, i32]),
            "gdgh_chain_set_numeric": (cs, [vp, i32, cs, C.c_int32]), "gdgh_chain_get_numeric": (cs, [vp, i32, cs, C.POINTER(C.c_int32)]),
            "gdgh_chain_length": (i32, [vp]), "gdgh_chain_parameters": (cs, [vp, i32, vp, i32]),
            "gdgh_chain_process": (None, [vp, vp, i32, vp, i32, C.c_uint32]),
            "gdgh_unit_create": (vp, [i32]), "gdgh_unit_destroy": (None, [vp]),
            "gdgh_unit_set_numeric": (cs, [vp, cs, C.c_int32]), "gdgh_unit_set_discrete": (cs, [vp, cs, cs]),
            "gdgh_unit_process": (None, [vp, vp, vp, i32, C.c_uint32]),
            "gdgh_filter_compile": (i32, [vp, i32, C.c_uint32, C.c_int32, C.c_uint32, C.c_int32, vp, i32]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _err(msg):
    if msg is not None:
        raise HostError(msg.decode())


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class ImpulseResponses:
    """filter.ImpulseResponses filled from memory (the reference imports WAV files, out of scope)."""

    def __init__(self):
        self._h = lib().gdgh_irs_create()

    def add(self, name, sample_rate, compensation_db, taps):
        t = _f64(taps)
        lib().gdgh_irs_add(self._h, name.encode(), sample_rate, compensation_db, t.ctypes.data, t.size)


class Engine:
    def __init__(self, n_channels, max_frames=8192, device=0, devices=None):
        """devices: one shard (context) per entry -- the same device may appear several times (independent contexts)."""
        if devices is None:
            self._h = lib().gdgh_engine_create(n_channels, max_frames, device)
        else:
            arr = (C.c_int * len(devices))(*devices)
            self._h = lib().gdgh_engine_create_sharded(n_channels, max_frames, arr, len(devices))
        self.n_channels = n_channels
        self.chains = []

    def shards(self):
        return lib().gdgh_engine_shards(self._h)

    def shard_of(self, channel):
        return lib().gdgh_engine_shard_of(self._h, channel)

    def close(self):
        if getattr(self, "_h", None):
            for c in self.chains:
                c._close()
            lib().gdgh_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_rendezvous(self, expected, timeout_ms):
        lib().gdgh_engine_set_rendezvous(self._h, expected, timeout_ms)

    def last_error(self):
        return lib().gdgh_engine_last_error(self._h).decode()

    def create_chain(self, irs=None):
        h = lib().gdgh_chain_create(self._h, irs._h if irs is not None else None)
        if not h:
            raise HostError("CreateChain failed")
        c = Chain(h, irs)
        self.chains.append(c)
        return c

    def process_all(self, x, sample_rate):
        x = _f64(x)
        n = x.shape[0]
        out = np.empty_like(x)
        ins = (C.c_void_p * n)(*[x[i].ctypes.data for i in range(n)])
        outs = (C.c_void_p * n)(*[out[i].ctypes.data for i in range(n)])
        _err(lib().gdgh_engine_process_all(self._h, ins, outs, x.shape[1], sample_rate))
        return out


    def shard_range(self, shard):
        first, count = C.c_int(0), C.c_int(0)
        lib().gdgh_engine_shard_range(self._h, shard, C.byref(first), C.byref(count))
        return first.value, count.value

    def raw_context(self, shard=0):
        """the shard's gdg_ctx handle (for configuring what the twin has no class for: metronome, meters) wrapped as a package Context
        that does NOT own it"""
        import __graft_entry__ as entry
        pkg = entry.load_package()
        ctx = pkg.Context.__new__(pkg.Context)
        ctx._h = C.c_void_p(lib().gdgh_engine_context(self._h, shard))
        ctx.n_channels = self.shard_range(shard)[1]
        ctx.max_frames, ctx.device, ctx._chains = 8192, 0, []
        ctx.close = lambda: None                      # the engine owns the context
        return ctx

    def batch_run(self, inputs, target_rate, out_format, window=16, metronome_to_master=False, run_meters=False, tuner_enqueue=False):
        """Engine::BatchRun: controller.processFiles' data path over all shards; returns the N + 3 output data sections."""
        import __graft_entry__ as entry
        pkg = entry.load_package()
        n = len(inputs)
        arr = (pkg.BatchInput * n)()
        keep = []
        for i, it in enumerate(inputs):
            if it is None:
                continue
            data, fmt, rate = it[0], it[1], it[2]
            channels, channel = (it[3], it[4]) if len(it) > 3 else (1, 0)
            f = pkg.WAVE_FORMATS[fmt] if isinstance(fmt, str) else fmt
            data = np.ascontiguousarray(data, dtype=np.uint8)
            keep.append(data)
            w = max(pkg.lib().gdg_wave_bytes_per_sample(f), 1)
            arr[i] = pkg.BatchInput(data.ctypes.data if data.size else None, data.size // (w * max(channels, 1)), f, rate, channels, channel)
        fo = pkg.WAVE_FORMATS[out_format] if isinstance(out_format, str) else out_format
        opt = pkg.BatchOptions(target_rate, fo, int(bool(metronome_to_master)), int(bool(run_meters)), int(bool(tuner_enqueue)))
        wo = pkg.lib().gdg_wave_bytes_per_sample(fo)
        # the job's length: the longest shard's (asked from the shards' own contexts)
        length = 0
        for g in range(self.shards()):
            first, count = self.shard_range(g)
            if count > 0:
                length = max(length, self.raw_context(g).batch_length(inputs[first:first + count], target_rate))
        outs = [np.zeros(length * wo, dtype=np.uint8) for _ in range(n + 3)]
        ptrs = (C.c_void_p * (n + 3))(*[(o.ctypes.data if o.size else None) for o in outs])
        samples = C.c_size_t(0)
        _err(lib().gdgh_engine_batch_run(self._h, arr, n, C.byref(opt), window, ptrs, C.byref(samples)))
        assert samples.value == length
        return outs

class Chain:
    """signal.Chain: the 14 methods of signal/signal.go:21-36."""

    def __init__(self, handle, irs):
        self._h = handle
        self._irs = irs          # keep the library alive

    def _close(self):
        if self._h:
            lib().gdgh_chain_destroy(self._h)
            self._h = None

    def AppendUnit(self, unit_type):
        i = C.c_int(-1)
        _err(lib().gdgh_chain_append_unit(self._h, unit_type, C.byref(i)))
        return i.value

    def RemoveUnit(self, i):
        _err(lib().gdgh_chain_remove_unit(self._h, i))

    def MoveUp(self, i):
        _err(lib().gdgh_chain_move_up(self._h, i))

    def MoveDown(self, i):
        _err(lib().gdgh_chain_move_down(self._h, i))

    def UnitType(self, i):
        t = C.c_int(-1)
        _err(lib().gdgh_chain_unit_type(self._h, i, C.byref(t)))
        return t.value

    def SetBypass(self, i, bypass):
        _err(lib().gdgh_chain_set_bypass(self._h, i, 1 if bypass else 0))

    def GetBypass(self, i):
        b = C.c_int(0)
        _err(lib().gdgh_chain_get_bypass(self._h, i, C.byref(b)))
        return bool(b.value)

    def SetDiscreteValue(self, i, name, value):
        _err(lib().gdgh_chain_set_discrete(self._h, i, name.encode(), value.encode()))

    def GetDiscreteValue(self, i, name):
        buf = C.create_string_buffer(512)
        _err(lib().gdgh_chain_get_discrete(self._h, i, name.encode(), buf, 512))
        return buf.value.decode()

    def SetNumericValue(self, i, name, value):
        _err(lib().gdgh_chain_set_numeric(self._h, i, name.encode(), int(value)))

    def GetNumericValue(self, i, name):
        v = C.c_int32(0)
        _err(lib().gdgh_chain_get_numeric(self._h, i, name.encode(), C.byref(v)))
        return v.value

    def Parameters(self, i):
        buf = C.create_string_buffer(1 << 16)
        _err(lib().gdgh_chain_parameters(self._h, i, buf, 1 << 16))
        out = []
        for line in buf.value.decode().splitlines():
            name, typ, unit, mn, mx, num, idx, vals = line.split("|")
            out.append({"Name": name, "Type": int(typ), "PhysicalUnit": unit, "Minimum": int(mn), "Maximum": int(mx),
                        "NumericValue": int(num), "DiscreteValueIndex": int(idx), "DiscreteValues": vals.split(";") if vals else []})
        return out

    def Length(self):
        return lib().gdgh_chain_length(self._h)

    def Process(self, x, sample_rate, n_out=None):
        x = _f64(x)
        n_out = x.size if n_out is None else n_out
        out = np.full(n_out, np.nan)
        lib().gdgh_chain_process(self._h, x.ctypes.data, x.size, out.ctypes.data, n_out, sample_rate)
        return out


class Spatializer:
    """spatializer.Spatializer: the ten methods of spatializer/spatializer.go:30-41 on top of an Engine's shards."""

    def __init__(self, engine, input_channels):
        self._engine = engine
        self._h = lib().gdgh_spatializer_create(engine._h, input_channels)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gdgh_spatializer_destroy(self._h)
            self._h = None

    def _get(self, what, ch):
        v = C.c_double(0.0)
        _err(lib().gdgh_spatializer_get(self._h, what, ch, C.byref(v)))
        return v.value

    def GetAzimuth(self, ch): return self._get(0, ch)
    def GetDistance(self, ch): return self._get(1, ch)
    def GetLevel(self, ch): return self._get(2, ch)
    def SetAzimuth(self, ch, v): _err(lib().gdgh_spatializer_set(self._h, 0, ch, float(v)))
    def SetDistance(self, ch, v): _err(lib().gdgh_spatializer_set(self._h, 1, ch, float(v)))
    def SetLevel(self, ch, v): _err(lib().gdgh_spatializer_set(self._h, 2, ch, float(v)))
    def GetInputCount(self): return lib().gdgh_spatializer_input_count(self._h)
    def GetOutputCount(self): return lib().gdgh_spatializer_output_count(self._h)
    def SetSampleRate(self, rate): lib().gdgh_spatializer_set_sample_rate(self._h, rate)

    def Process(self, x, aux=None, reuse_chain_outputs=False):
        x = _f64(x)
        n = x.shape[1]
        ins = (C.c_void_p * x.shape[0])(*[x[i].ctypes.data for i in range(x.shape[0])])
        a = _f64(aux) if aux is not None else None
        left, right = np.full(n, np.nan), np.full(n, np.nan)
        lib().gdgh_spatializer_process(self._h, ins, a.ctypes.data if a is not None else None, left.ctypes.data, right.ctypes.data, n,
                                       1 if reuse_chain_outputs else 0)
        return left, right


class Tuner:
    """tuner.Tuner (tuner/tuner.go:62-65)."""

    def __init__(self, device=0):
        self._h = lib().gdgh_tuner_create(device)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gdgh_tuner_destroy(self._h)
            self._h = None

    def Process(self, samples, sample_rate):
        s = _f64(samples)
        lib().gdgh_tuner_process(self._h, s.ctypes.data, s.size, sample_rate)

    def Analyze(self):
        cents, freq, note = C.c_int(0), C.c_double(0.0), C.create_string_buffer(64)
        _err(lib().gdgh_tuner_analyze(self._h, C.byref(cents), C.byref(freq), note, 64))
        return {"cents": cents.value, "frequency": freq.value, "note": note.value.decode()}


class Unit:
    """A stand-alone effects.Unit (runs on a private one-channel context)."""

    def __init__(self, unit_type):
        self._h = lib().gdgh_unit_create(unit_type)
        if not self._h:
            raise HostError("Failed to create effects unit.")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gdgh_unit_destroy(self._h)
            self._h = None

    def SetNumericValue(self, name, value):
        _err(lib().gdgh_unit_set_numeric(self._h, name.encode(), int(value)))

    def SetDiscreteValue(self, name, value):
        _err(lib().gdgh_unit_set_discrete(self._h, name.encode(), value.encode()))

    def Process(self, x, sample_rate):
        x = _f64(x)
        out = np.empty_like(x)
        lib().gdgh_unit_process(self._h, x.ctypes.data, out.ctypes.data, x.size, sample_rate)
        return out


def filter_compile(taps, sample_rate, compensation_db, order, level_db):
    """Reduce(order) -> Normalize -> Multiply(level) of one IR, as poweramp.compile does per slot."""
    t = _f64(taps)
    cap = max(t.size, order if order else 0) + 8
    out = np.zeros(cap)
    n = lib().gdgh_filter_compile(t.ctypes.data, t.size, sample_rate, compensation_db, order, level_db, out.ctypes.data, cap)
    return out[:n]
