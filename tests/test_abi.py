"""The C-ABI library loads on a CPU-only box and exports every symbol include/*.h declares;
struct layouts match the reference's."""
import ctypes as C
import os
import re

import bliss_amd
from bliss_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(bl_[a-z0-9_]+)\s*\(", text))


def test_every_declared_symbol_is_exported(lib):
    names = _declared("bliss.h") | _declared("bliss_amd.h")
    assert len(names) >= 28
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"
    assert names == set(_lib.SYMBOLS), names ^ set(_lib.SYMBOLS)


def test_struct_layout_matches_reference():
    # ref include/bliss.h:49-67 on x86-64: 120 bytes, offsets from SURVEY.md §8b
    S = _lib.BlSong
    assert C.sizeof(S) == 120
    want = dict(force=0, force_vector=4, sample_array=24, channels=32, nSamples=36,
                sample_rate=40, bitrate=44, nb_bytes_per_sample=48, calm_or_loud=52,
                resampled=56, duration=64, filename=72, artist=80, title=88, album=96,
                tracknumber=104, genre=112)
    for k, off in want.items():
        assert getattr(S, k).offset == off, k
    assert C.sizeof(_lib.ForceVector) == 16
    assert [f[0] for f in _lib.ForceVector._fields_] == ["tempo", "amplitude", "frequency", "attack"]
    assert C.sizeof(_lib.SongDesc) == 24 and C.sizeof(_lib.SongResult) == 80


def test_no_device_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        return
    assert lib.bl_amd_device_count() == 0
    assert lib.bl_amd_init(0) == _lib.BL_UNEXPECTED
    a = _lib.ForceVector(1, 2, 3, 4)
    assert lib.bl_distance(a, a) == float(_lib.BL_UNEXPECTED)  # no CPU path behind it


def test_return_codes():
    assert (bliss_amd.BL_LOUD, bliss_amd.BL_CALM, bliss_amd.BL_UNKNOWN,
            bliss_amd.BL_UNEXPECTED, bliss_amd.BL_OK) == (0, 1, 2, -2, 0)
