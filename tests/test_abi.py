"""The C-ABI library loads on a CPU-only box and exports every symbol include/*.h declares;
struct layouts match the reference's."""
import ctypes as C
import os
import re

import pytest

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


def test_nothing_else_is_exported(lib):
    """`nm -D`: the dynamic symbol table holds the 15 symbols of the reference's libbliss.so (ref include/bliss.h:80-290)
    and the bl_amd_* extension — no kernels, launch-layer functions, decoder internals or libstdc++ instantiations
    (built with -fvisibility=hidden and the export list bliss_amd/csrc/libbliss_amd.map)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], check=True, stdout=subprocess.PIPE, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    declared = _declared("bliss.h") | _declared("bliss_amd.h")
    assert exported == declared, sorted(exported ^ declared)
    assert len([n for n in exported if not n.startswith("bl_amd_")]) == 15


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
    """No CPU path behind the analysis: every analysis entry point reports BL_UNEXPECTED."""
    import numpy as np
    import torch
    if torch.cuda.is_available():
        return
    assert lib.bl_amd_device_count() == 0
    assert lib.bl_amd_init(0) == _lib.BL_UNEXPECTED
    pcm = np.ones(8192, dtype=np.int16)
    song = _lib.BlSong()
    song.sample_array = pcm.ctypes.data
    song.nSamples, song.channels, song.duration = pcm.size, 1, 1
    assert lib.bl_amplitude_sort(C.byref(song)) == float(_lib.BL_UNEXPECTED)
    assert lib.bl_frequency_sort(C.byref(song)) == float(_lib.BL_UNEXPECTED)
    env = _lib.EnvelopeResult()
    lib.bl_envelope_sort(C.byref(song), C.byref(env))
    assert env.tempo == float(_lib.BL_UNEXPECTED)
    assert lib.bl_mean(pcm.ctypes.data_as(C.POINTER(C.c_int16)), pcm.size) == _lib.BL_UNEXPECTED
    s2 = _lib.BlSong()
    flac = os.path.join(ROOT, "tests", "golden", "song.flac").encode()
    assert lib.bl_analyze(flac, C.byref(s2)) == _lib.BL_UNEXPECTED   # decodes, then no device
    lib.bl_free_song(C.byref(s2))
    with __import__("pytest").raises(RuntimeError):
        bliss_amd.analyze_batch_host([pcm], 1, 1)
    ctx = C.c_void_p()
    assert lib.bl_amd_ctx_create(0, C.byref(ctx)) == _lib.BL_UNEXPECTED and not ctx.value


def test_analyze_files_fails_loudly_without_a_device(lib, tmp_path):
    """The file-batch call has no CPU path either: without a HIP device it returns BL_UNEXPECTED."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    names = (C.c_char_p * 1)(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "song.flac").encode())
    songs = (_lib.BlSong * 1)()
    codes = (C.c_int * 1)()
    assert lib.bl_amd_analyze_files(names, 1, songs, codes, 1, 0) == _lib.BL_UNEXPECTED
    assert lib.bl_amd_analyze_files(None, 1, songs, codes, 1, 0) == _lib.BL_UNEXPECTED


def test_scalar_helpers_are_the_reference_expressions(lib, oracle):
    """bl_distance / bl_cosine_similarity of one pair and bl_rectangular_filter are host
    arithmetic (bl_api.c); bit-identical to the oracle here, and to the GPU's all-pairs kernels
    in tests/test_gpu_parity.py."""
    import numpy as np
    rng = np.random.default_rng(8)
    v = (rng.standard_normal((64, 4)) * 10).astype(np.float32)
    for i in range(0, 64, 2):
        a, b = _lib.ForceVector(*v[i]), _lib.ForceVector(*v[i + 1])
        assert lib.bl_distance(a, b) == oracle.distance(v[i], v[i + 1])
        assert lib.bl_cosine_similarity(a, b) == oracle.cosine(v[i], v[i + 1])
    dp = C.POINTER(C.c_double)
    for n, w in ((500, 19), (20, 19), (40, 19), (64, 7), (40, 4), (64, 8), (33, 2), (19, 19), (8, 8)):  # even widths too
        inp, old = rng.standard_normal(n), rng.standard_normal(n)
        out = old.copy()
        lib.bl_rectangular_filter(out.ctypes.data_as(dp), inp.ctypes.data_as(dp), n, w)
        assert np.array_equal(out, oracle.rect_filter(old, inp, w)), (n, w)


def test_return_codes():
    assert (bliss_amd.BL_LOUD, bliss_amd.BL_CALM, bliss_amd.BL_UNKNOWN,
            bliss_amd.BL_UNEXPECTED, bliss_amd.BL_OK) == (0, 1, 2, -2, 0)


def test_headers_compile_as_c99_without_hip():
    """include/bliss.h and include/bliss_amd.h are plain C: the resident-corpus caller of
    INTEGRATION.md section 4 compiles with gcc -std=c99 and nothing but the two headers."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for src in ("multi_device_caller.c", "dropin_check.c", "caller_relying_on_transitive_headers.c"):
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only",
                        "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c", src)], check=True)
