"""ctypes access to oracle/liboracle.so — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")


class OrcResult(C.Structure):  # oracle/bliss_oracle.h orc_result
    _fields_ = [("tempo", C.c_float), ("amplitude", C.c_float), ("frequency", C.c_float),
                ("attack", C.c_float), ("force", C.c_float), ("calm_or_loud", C.c_int),
                ("start", C.c_int), ("end", C.c_int), ("mean", C.c_int), ("variance", C.c_int),
                ("n_frames", C.c_int), ("nb_frames", C.c_int), ("n_windows", C.c_int),
                ("beat", C.c_int), ("atk_sum", C.c_double), ("min_peak_margin", C.c_double),
                ("hist_integral", C.c_float), ("freq_peak", C.c_float)]


FIELDS = [f[0] for f in OrcResult._fields_]


def build_oracle():
    if not os.path.exists(LIB) or any(
            os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(LIB)
            for f in ("bliss_oracle.c", "orc_fft.c", "orc_fft_alt.c", "orc_fft_lavc.c", "orc_synth.c", "bliss_oracle.h")):
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)
    return LIB


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(build_oracle())
        L = self.lib
        i16p = C.POINTER(C.c_int16)
        L.orc_analyze_pcm.restype = C.c_int
        L.orc_analyze_pcm.argtypes = [i16p, C.c_int, C.c_int, C.c_uint64, C.POINTER(OrcResult)]
        L.orc_envelope.restype = None
        L.orc_envelope.argtypes = [i16p, C.c_int, C.c_uint64, C.POINTER(OrcResult), C.POINTER(C.c_float)]
        L.orc_amplitude.restype = C.c_float
        L.orc_amplitude.argtypes = [i16p, C.c_int, C.POINTER(OrcResult)]
        L.orc_frequency.restype = C.c_float
        L.orc_frequency.argtypes = [i16p, C.c_int, C.c_int, C.POINTER(OrcResult)]
        L.orc_mean.restype = C.c_int
        L.orc_mean.argtypes = [i16p, C.c_int]
        L.orc_variance.restype = C.c_int
        L.orc_variance.argtypes = [i16p, C.c_int, C.c_int]
        L.orc_rect_filter.restype = None
        L.orc_rect_filter.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_int]
        L.orc_distance.restype = C.c_float
        L.orc_distance.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_cosine.restype = C.c_float
        L.orc_cosine.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_distance_matrix.restype = None
        L.orc_distance_matrix.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float)]
        L.orc_cosine_matrix.restype = None
        L.orc_cosine_matrix.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float)]
        L.orc_synth_fill.restype = None
        L.orc_synth_fill.argtypes = [i16p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_set_fft_variant.restype = None
        L.orc_set_fft_variant.argtypes = [C.c_int]

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.POINTER(C.c_int16))

    def set_fft_variant(self, v):
        """0 the defaults (f64: packed radix-2; f32: libavcodec's operation order, orc_fft_lavc.c), 1 recursive radix-4 on the
        unpacked input, 2 the defining sum (orc_fft_alt.c), 3 = the f32 default by name, 4 the f32 packed radix-2"""
        self.lib.orc_set_fft_variant(v)

    def frequency(self, pcm, channels):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        return float(self.lib.orc_frequency(self._p(pcm), pcm.size, channels, None))

    def synth(self, seed, rate, channels, n):
        out = np.empty(n, dtype=np.int16)
        self.lib.orc_synth_fill(self._p(out), n, seed, rate, channels)
        return out

    def analyze(self, pcm, channels, duration):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        r = OrcResult()
        self.lib.orc_analyze_pcm(self._p(pcm), pcm.size, channels, duration, C.byref(r))
        return {k: getattr(r, k) for k in FIELDS}

    def envelope(self, pcm, duration):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        r = OrcResult()
        en = np.zeros(2 * (pcm.size // 512) + 4, dtype=np.float32)
        self.lib.orc_envelope(self._p(pcm), pcm.size, duration, C.byref(r),
                              en.ctypes.data_as(C.POINTER(C.c_float)))
        return {k: getattr(r, k) for k in FIELDS}, en

    def mean(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        return self.lib.orc_mean(self._p(pcm), pcm.size)

    def variance(self, pcm, mean):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        return self.lib.orc_variance(self._p(pcm), pcm.size, mean)

    def rect_filter(self, out, inp, width):
        out = np.ascontiguousarray(out, dtype=np.float64).copy()
        inp = np.ascontiguousarray(inp, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        self.lib.orc_rect_filter(out.ctypes.data_as(dp), inp.ctypes.data_as(dp), out.size, width)
        return out

    def distance_matrix(self, vecs):
        v = np.ascontiguousarray(vecs, dtype=np.float32).reshape(-1, 4)
        out = np.empty((v.shape[0], v.shape[0]), dtype=np.float32)
        fp = C.POINTER(C.c_float)
        self.lib.orc_distance_matrix(v.ctypes.data_as(fp), v.shape[0], out.ctypes.data_as(fp))
        return out

    def cosine_matrix(self, vecs):
        v = np.ascontiguousarray(vecs, dtype=np.float32).reshape(-1, 4)
        out = np.empty((v.shape[0], v.shape[0]), dtype=np.float32)
        fp = C.POINTER(C.c_float)
        self.lib.orc_cosine_matrix(v.ctypes.data_as(fp), v.shape[0], out.ctypes.data_as(fp))
        return out

    def cosine(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        fp = C.POINTER(C.c_float)
        return self.lib.orc_cosine(a.ctypes.data_as(fp), b.ctypes.data_as(fp))

    def distance(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        fp = C.POINTER(C.c_float)
        return self.lib.orc_distance(a.ctypes.data_as(fp), b.ctypes.data_as(fp))
