"""Committed golden vectors (tests/golden/synth_golden.json, made by
tests/golden/make_synth_golden.py from the oracle): the oracle must keep reproducing them on
every machine (CPU test), and the HIP path must match them (GPU test) — integers and window
energies bit-exact, f32 features to 1e-4 relative."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "synth_golden.json")))
INTS = ("start", "end", "mean", "variance", "n_frames", "nb_frames", "n_windows", "beat", "calm_or_loud")
FLOATS = ("tempo", "amplitude", "frequency", "attack", "force")


def _pcm(oracle, c):
    pcm = oracle.synth(c["seed"], c["rate"], c["channels"], c["n_samples"])
    assert hashlib.md5(pcm.tobytes()).hexdigest() == c["pcm_md5"], "generator drifted"
    return pcm


def test_oracle_matches_committed_golden(oracle):
    for c in GOLD["cases"]:
        pcm = _pcm(oracle, c)
        r = oracle.analyze(pcm, c["channels"], c["duration"])
        for k in INTS:
            assert int(r[k]) == int(c["expect"][k]), (c["seed"], k)
        for k in FLOATS:  # same code, possibly another libm: allow the last bits of log/log10/cos
            assert abs(r[k] - c["expect"][k]) <= 1e-6 * max(1.0, abs(c["expect"][k])), (c["seed"], k)
        _, en = oracle.envelope(pcm, c["duration"])
        assert hashlib.md5(en[:r["n_windows"]].tobytes()).hexdigest() == c["energies_md5"], c["seed"]


@pytest.mark.gpu
def test_hip_path_matches_committed_golden(gpu_lib, oracle):
    import bliss_amd
    cases = GOLD["cases"]
    corpus = bliss_amd.DeviceCorpus([c["n_samples"] for c in cases], [c["channels"] for c in cases],
                                    [c["duration"] for c in cases])
    for i, c in enumerate(cases):
        corpus.upload(i, _pcm(oracle, c))
    corpus.analyze()
    got = corpus.fetch()
    total = int(sum(int(g["nb_frames"]) for g in got))
    en = np.zeros(total, dtype=np.float32)
    assert gpu_lib.bl_amd_last_energies(en.ctypes.data_as(C.POINTER(C.c_float)), total) == total
    # the committed energies are the reference arithmetic's: FIR mode 0 reproduces them bit for bit;
    # the default mode (bl_amd_set_fir_mode) may sit one f32 ulp off in about one window per 10^8
    try:
        assert gpu_lib.bl_amd_set_fir_mode(0) == 0
        corpus.analyze()
        got0 = corpus.fetch()
        en0 = np.zeros(total, dtype=np.float32)
        assert gpu_lib.bl_amd_last_energies(en0.ctypes.data_as(C.POINTER(C.c_float)), total) == total
    finally:
        gpu_lib.bl_amd_set_fir_mode(-1)
    off = 0
    moved = 0
    for i, c in enumerate(cases):
        for k in INTS:
            assert int(got[i][k]) == int(c["expect"][k]) == int(got0[i][k]), (c["seed"], k, int(got[i][k]))
        for k in FLOATS:
            a, b = float(got[i][k]), float(c["expect"][k])
            assert abs(a - b) <= 1e-4 * max(abs(b), 1e-6), (c["seed"], k, a, b)
        nw = int(got[i]["n_windows"])
        assert hashlib.md5(en0[off:off + nw].tobytes()).hexdigest() == c["energies_md5"], (c["seed"], "window energies")
        d = np.abs(en[off:off + nw].view(np.int32).astype(np.int64) - en0[off:off + nw].view(np.int32).astype(np.int64))
        assert d.max() <= 1, (c["seed"], "default FIR mode more than one ulp off")
        moved += int(np.count_nonzero(d))
        off += int(got[i]["nb_frames"])
    assert moved <= 2, moved
