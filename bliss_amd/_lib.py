"""ctypes binding of libbliss_amd.so (the C-ABI of include/bliss.h + include/bliss_amd.h).

This mirrors what the reference's cffi module `bliss._bliss` exposes
(ref python/build_bliss.py:35-38: cdef = include/bliss.h minus '#' lines) plus the
batch extension.  The shared object is built in-tree by `__graft_entry__.build()`
(make -C bliss_amd/csrc) and loaded from this directory; there is no fallback:
a missing library raises ImportError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BLISS_AMD_LIB: load another build of the same C-ABI (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("BLISS_AMD_LIB") or os.path.join(_HERE, "libbliss_amd.so")

BL_LOUD, BL_CALM, BL_UNKNOWN, BL_UNEXPECTED, BL_OK = 0, 1, 2, -2, 0


class ForceVector(C.Structure):  # ref include/bliss.h:26-31
    _fields_ = [("tempo", C.c_float), ("amplitude", C.c_float),
                ("frequency", C.c_float), ("attack", C.c_float)]


class EnvelopeResult(C.Structure):  # ref include/bliss.h:34-37
    _fields_ = [("tempo", C.c_float), ("attack", C.c_float)]


class BlSong(C.Structure):  # ref include/bliss.h:49-67
    _fields_ = [("force", C.c_float), ("force_vector", ForceVector),
                ("sample_array", C.c_void_p), ("channels", C.c_int), ("nSamples", C.c_int),
                ("sample_rate", C.c_int), ("bitrate", C.c_int),
                ("nb_bytes_per_sample", C.c_int), ("calm_or_loud", C.c_int),
                ("resampled", C.c_int), ("duration", C.c_uint64),
                ("filename", C.c_char_p), ("artist", C.c_char_p), ("title", C.c_char_p),
                ("album", C.c_char_p), ("tracknumber", C.c_char_p), ("genre", C.c_char_p)]


class SongDesc(C.Structure):  # include/bliss_amd.h bl_amd_song_desc
    _fields_ = [("pcm_offset", C.c_uint64), ("n_samples", C.c_int32),
                ("channels", C.c_int32), ("duration", C.c_uint64)]


class ResampleDesc(C.Structure):  # include/bliss_amd.h bl_amd_resample_desc
    _fields_ = [("in_offset", C.c_uint64), ("out_offset", C.c_uint64), ("frames", C.c_int32),
                ("channels", C.c_int32)]


class SongResult(C.Structure):  # include/bliss_amd.h bl_amd_song_result
    _fields_ = [("v", ForceVector), ("force", C.c_float), ("calm_or_loud", C.c_int32),
                ("status", C.c_int32), ("start", C.c_int32), ("end", C.c_int32),
                ("mean", C.c_int32), ("variance", C.c_int32), ("n_frames", C.c_int32),
                ("nb_frames", C.c_int32), ("n_windows", C.c_int32), ("beat", C.c_int32),
                ("hist_integral", C.c_float), ("freq_peak", C.c_float),
                ("atk_sum", C.c_double)]


class Shard(C.Structure):  # include/bliss_amd.h bl_amd_shard
    _fields_ = [("device", C.c_int32), ("n_songs", C.c_int32), ("d_pcm", C.c_void_p),
                ("h_desc", C.POINTER(SongDesc)), ("d_results", C.c_void_p), ("d_rows", C.c_void_p)]


# every symbol include/*.h declares: name -> (restype, argtypes)
_P = C.POINTER
SYMBOLS = {
    # include/bliss.h
    "bl_analyze": (C.c_int, [C.c_char_p, _P(BlSong)]),
    "bl_distance_file": (C.c_float, [C.c_char_p, C.c_char_p, _P(BlSong), _P(BlSong)]),
    "bl_distance": (C.c_float, [ForceVector, ForceVector]),
    "bl_cosine_similarity_file": (C.c_float, [C.c_char_p, C.c_char_p, _P(BlSong), _P(BlSong)]),
    "bl_cosine_similarity": (C.c_float, [ForceVector, ForceVector]),
    "bl_envelope_sort": (None, [_P(BlSong), _P(EnvelopeResult)]),
    "bl_amplitude_sort": (C.c_float, [_P(BlSong)]),
    "bl_frequency_sort": (C.c_float, [_P(BlSong)]),
    "bl_audio_decode": (C.c_int, [C.c_char_p, _P(BlSong)]),
    "bl_free_song": (None, [_P(BlSong)]),
    "bl_version": (C.c_float, []),
    "bl_initialize_song": (None, [_P(BlSong)]),
    "bl_mean": (C.c_int, [_P(C.c_int16), C.c_int]),
    "bl_variance": (C.c_int, [_P(C.c_int16), C.c_int, C.c_int]),
    "bl_rectangular_filter": (None, [_P(C.c_double), _P(C.c_double), C.c_int, C.c_int]),
    # include/bliss_amd.h
    "bl_amd_init": (C.c_int, [C.c_int]),
    "bl_amd_device_count": (C.c_int, []),
    "bl_amd_ctx_create": (C.c_int, [C.c_int, _P(C.c_void_p)]),
    "bl_amd_ctx_destroy": (None, [C.c_void_p]),
    "bl_amd_ctx_device": (C.c_int, [C.c_void_p]),
    "bl_amd_analyze_batch_device": (C.c_int, [C.c_void_p, _P(SongDesc), C.c_int, C.c_void_p, C.c_void_p]),
    "bl_amd_ctx_analyze_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, _P(SongDesc), C.c_int, C.c_void_p,
                                                  C.c_void_p]),
    "bl_amd_ctx_analyze_batch_host": (C.c_int, [C.c_void_p, _P(C.c_void_p), _P(C.c_int32), _P(C.c_int32),
                                                _P(C.c_uint64), C.c_int, _P(SongResult)]),
    "bl_amd_analyze_files": (C.c_int, [_P(C.c_char_p), C.c_int, _P(BlSong), _P(C.c_int), C.c_int, C.c_int]),
    "bl_amd_set_host_transfer": (C.c_int, [C.c_int]),
    "bl_amd_analyze_batch_host_s32": (C.c_int, [_P(C.c_void_p), _P(C.c_int32), _P(C.c_int32),
                                                _P(C.c_uint64), C.c_int, _P(SongResult)]),
    "bl_amd_analyze_batch_host_rate": (C.c_int, [_P(C.c_void_p), C.c_int, _P(C.c_int32), _P(C.c_int32),
                                                 _P(C.c_uint64), C.c_int, C.c_int, _P(SongResult)]),
    "bl_amd_narrow_s32_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bl_amd_resample_out_frames": (C.c_size_t, [C.c_size_t, C.c_int]),
    "bl_amd_resample_host": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int,
                                       _P(_P(C.c_int16)), _P(C.c_size_t)]),
    "bl_amd_resample_batch_device": (C.c_int, [C.c_void_p, C.c_int, _P(ResampleDesc), C.c_int, C.c_int,
                                               C.c_void_p, C.c_void_p]),
    "bl_amd_ctx_resample_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _P(ResampleDesc), C.c_int,
                                                   C.c_int, C.c_void_p, C.c_void_p]),
    "bl_amd_analyze_corpus_multi": (C.c_int, [_P(C.c_void_p), _P(C.c_int32), _P(C.c_int32), _P(C.c_uint64),
                                              C.c_int, _P(C.c_int), C.c_int, C.c_int, _P(SongResult),
                                              _P(C.c_float)]),
    "bl_amd_analyze_corpus_multi_device": (C.c_int, [_P(Shard), C.c_int, C.c_int, _P(SongResult), _P(C.c_float)]),
    "bl_amd_decode_allow_native_rate": (None, [C.c_int]),
    "bl_amd_flac_verify": (C.c_int, [C.c_char_p, _P(C.c_uint8), _P(C.c_uint8)]),
    "bl_amd_analyze_batch_host": (C.c_int, [_P(C.c_void_p), _P(C.c_int32), _P(C.c_int32),
                                            _P(C.c_uint64), C.c_int, _P(SongResult)]),
    "bl_amd_distance_matrix_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bl_amd_cosine_matrix_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bl_amd_distance_matrix_host": (C.c_int, [_P(ForceVector), C.c_int, _P(C.c_float)]),
    "bl_amd_cosine_matrix_host": (C.c_int, [_P(ForceVector), C.c_int, _P(C.c_float)]),
    "bl_amd_selftest_sqrt": (C.c_int, [_P(C.c_uint64)]),
    "bl_amd_selftest_cos": (C.c_int, [_P(C.c_uint64), C.c_uint64]),
    "bl_amd_playlist_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bl_amd_playlist_host": (C.c_int, [_P(ForceVector), C.c_int, C.c_int, _P(C.c_int32), _P(C.c_float)]),
    "bl_amd_synth_pcm_device": (C.c_int, [C.c_void_p, _P(SongDesc), C.c_int, C.c_uint32, C.c_uint32, C.c_void_p]),
    "bl_amd_set_fir_mode": (C.c_int, [C.c_int]),
    "bl_amd_fir_mode": (C.c_int, []),
    "bl_amd_profile": (None, [C.c_int]),
    "bl_amd_profile_reset": (None, []),
    "bl_amd_profile_ms": (C.c_double, [C.c_char_p, _P(C.c_int)]),
    "bl_amd_last_energies": (C.c_longlong, [_P(C.c_float), C.c_longlong]),
    "bl_amd_shutdown": (None, []),
}

_lib = None


def load():
    """Load libbliss_amd.so (once) and attach prototypes.  Raises ImportError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C bliss_amd/csrc). bliss_amd has no pure-Python or CPU fallback.")
    try:  # share torch's HIP runtime (same SONAME) when torch is in the process
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C-ABI
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError -> missing export
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
