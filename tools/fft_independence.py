#!/usr/bin/env python3
"""How much of the oracle's results is owed to its choice of FFT (CPU only, test infrastructure).  The reference calls
FFTW3 (f64, window energies) and libavcodec's RDFT (f32, frequency rating); neither can run here, the oracle restates
them as a packed radix-2 (oracle/orc_fft.c; since round 6 the f32 one in libavcodec's own operation order,
oracle/orc_fft_lavc.c, which tests/test_fft_independence.py singles out).  This runs every case — the reference's recording tests/golden/song.flac
and the eight cases of tests/golden/synth_golden.json — under the three implementations of oracle/orc_fft_alt.c and
reports, per case and per pair of implementations: how many f32 window energies differ and by how many ulp, whether
any integer (beat, ...) or any of tempo / attack moves, and the absolute spread of `frequency`.
usage: python tools/fft_independence.py [--skip-direct-above 20000] [--out profiles/r05_fft_independence.json]"""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = {0: "the defaults: f64 packed radix-2 (orc_fft.c), f32 libavcodec's operation order since round 6 (orc_fft_lavc.c)", 1: "recursive radix-4, unpacked", 2: "defining sum, extended precision"}
INTS = ("start", "end", "mean", "variance", "n_frames", "nb_frames", "n_windows", "beat", "calm_or_loud")


def load_cases():
    from tests.oracle_py import Oracle
    import bliss_amd
    from bliss_amd import _lib
    orc = Oracle()
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "synth_golden.json")))
    lib = bliss_amd.load()
    song = _lib.BlSong()
    assert lib.bl_audio_decode(os.path.join(ROOT, "tests", "golden", "song.flac").encode(), C.byref(song)) == _lib.BL_OK
    pcm = np.ctypeslib.as_array(C.cast(song.sample_array, C.POINTER(C.c_int16)), shape=(song.nSamples,)).copy()
    lib.bl_free_song(C.byref(song))
    cases = [dict(name="song.flac (ref audio/song.flac)", pcm=pcm, channels=2, duration=11)]
    for c in gold["cases"]:
        cases.append(dict(name=f"synth seed {c['seed']} ({c['rate']} Hz, {c['channels']} ch, {c['duration']} s)",
                          pcm=orc.synth(c["seed"], c["rate"], c["channels"], c["n_samples"]), channels=c["channels"],
                          duration=c["duration"]))
    return cases


def extra_cases(n, seconds=60):
    """n more synthetic songs (seeds 5000 ...), radix-2 against radix-4 only: the statistics behind 'an f32 rounding of
    the ordered sum flips about once per 10^9 additions'"""
    from tests.oracle_py import Oracle
    orc = Oracle()
    return [dict(name=f"extra seed {5000 + i} (44100 Hz, 2 ch, {seconds} s)", pcm=orc.synth(5000 + i, 44100, 2, 44100 * 2 * seconds),
                 channels=2, duration=seconds, extra=True) for i in range(n)]


def run(job):
    case, variant = job
    from tests.oracle_py import Oracle
    orc = Oracle()
    orc.set_fft_variant(variant)
    t0 = time.time()
    r = orc.analyze(case["pcm"], case["channels"], case["duration"])
    _, en = orc.envelope(case["pcm"], case["duration"])
    orc.set_fft_variant(0)
    return case["name"], variant, r, en[:r["n_windows"]].copy(), time.time() - t0


def compare(a, b):
    (ra, ea), (rb, eb) = a, b
    d = np.abs(ea.view(np.int32).astype(np.int64) - eb.view(np.int32).astype(np.int64))
    f32 = lambda x: np.float32(x).view(np.int32)
    return {"windows": int(ea.size), "energies_differing": int(np.count_nonzero(d)), "max_ulp": int(d.max()) if d.size else 0,
            "integers_identical": all(int(ra[k]) == int(rb[k]) for k in INTS),
            "tempo_attack_bit_identical": bool(f32(ra["tempo"]) == f32(rb["tempo"]) and f32(ra["attack"]) == f32(rb["attack"])),
            "amplitude_bit_identical": bool(f32(ra["amplitude"]) == f32(rb["amplitude"])),
            "frequency_abs_diff": float(abs(np.float64(ra["frequency"]) - np.float64(rb["frequency"]))),
            "frequency": [float(ra["frequency"]), float(rb["frequency"])]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-direct-above", type=int, default=10 ** 9, help="no defining-sum run for cases with more windows")
    ap.add_argument("--out", default="")
    ap.add_argument("--procs", type=int, default=max(1, (os.cpu_count() or 2) - 1))
    ap.add_argument("--extra", type=int, default=0, help="this many more one-minute synthetic songs, implementations 0 and 1 only")
    a = ap.parse_args()
    cases = load_cases() + extra_cases(a.extra)
    jobs = [(c, v) for c in cases for v in (0, 1, 2)
            if v < 2 or (2 * (c["pcm"].size // 512) <= a.skip_direct_above and not c.get("extra"))]
    jobs.sort(key=lambda j: -j[0]["pcm"].size * (40 if j[1] == 2 else 1))
    with mp.Pool(a.procs) as pool:
        res = pool.map(run, jobs, chunksize=1)
    by = {}
    for name, v, r, en, dt in res:
        by.setdefault(name, {})[v] = (r, en, dt)
    out = {"implementations": NAMES, "cases": {}, "totals": {}}
    tot = {"windows": 0, "0 vs 1": 0, "0 vs 2": 0, "1 vs 2": 0, "windows_with_direct": 0, "max_frequency_abs_diff": 0.0}
    for c in cases:
        e = by[c["name"]]
        rep = {"seconds": {NAMES[v]: round(e[v][2], 2) for v in e}}
        for x, y in ((0, 1), (0, 2), (1, 2)):
            if x in e and y in e:
                rep[f"{x} vs {y}"] = compare(e[x][:2], e[y][:2])
                tot[f"{x} vs {y}"] += rep[f"{x} vs {y}"]["energies_differing"]
                tot["max_frequency_abs_diff"] = max(tot["max_frequency_abs_diff"], rep[f"{x} vs {y}"]["frequency_abs_diff"])
        tot["windows"] += int(e[0][1].size)
        if 2 in e:
            tot["windows_with_direct"] += int(e[0][1].size)
        if c.get("extra"):  # the extra songs only enter the totals (and the list of differences, if any)
            if rep["0 vs 1"]["energies_differing"] or not rep["0 vs 1"]["integers_identical"]:
                out["cases"][c["name"]] = rep
            tot["extra_songs"] = tot.get("extra_songs", 0) + 1
            continue
        out["cases"][c["name"]] = rep
    out["totals"] = tot
    txt = json.dumps(out, indent=1)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
