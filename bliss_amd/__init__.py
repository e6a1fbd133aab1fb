"""bliss_amd — MI355X-native implementation of the bliss per-song analysis hot path.

Python surface mirrors the reference's `bliss` package (ref python/bliss/__init__.py:7-12)
for the analysis path and adds the batch entry points of include/bliss_amd.h.
All computation happens in libbliss_amd.so (hand-written HIP kernels for gfx950).
"""
from . import _lib
from ._lib import (BL_CALM, BL_LOUD, BL_OK, BL_UNEXPECTED, BL_UNKNOWN, BlSong, EnvelopeResult,
                   ForceVector, SongDesc, SongResult, load)
from . import distance, version
from .batch import (Context, DeviceCorpus, analyze_batch_host, analyze_files, analyze_batch_host_rate, analyze_batch_host_s32,
                    analyze_corpus_multi, analyze_corpus_multi_device, cosine_matrix, distance_matrix, playlist, resample_batch_device,
                    resample_host, results_to_numpy)
from .bl_song import bl_song

__all__ = ["_lib", "load", "BlSong", "ForceVector", "EnvelopeResult", "SongDesc", "SongResult",
           "BL_LOUD", "BL_CALM", "BL_UNKNOWN", "BL_UNEXPECTED", "BL_OK", "DeviceCorpus",
           "analyze_batch_host", "analyze_files", "analyze_batch_host_rate", "analyze_batch_host_s32", "analyze_corpus_multi", "analyze_corpus_multi_device", "Context",
           "distance_matrix", "cosine_matrix", "results_to_numpy", "playlist",
           "resample_host", "resample_batch_device", "bl_song", "distance", "version"]


def __getattr__(name):
    """`bliss_amd.lib`: the loaded C library, the counterpart of the reference's `bliss.lib`
    (ref python/bliss/__init__.py:5) — e.g. `bliss_amd.lib.bl_version()`.  Loaded on first use."""
    if name == "lib":
        return load()
    raise AttributeError(f"module 'bliss_amd' has no attribute {name!r}")
