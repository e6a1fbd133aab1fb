#!/usr/bin/env python3
"""Randomised parity soak: N random songs (length, channels, level, spectrum, DC, silences) through the
HIP batch path and through the CPU oracle (one process per host core), integers compared exactly,
tempo / amplitude / attack to 1e-4 relative (north-star) and frequency / force with
|x - y| <= 1e-5 + 1e-4 |y| (plus the absolute 1e-5 of the reference's own test, ref
tests/test_analyze.c:30-35: they are differences of O(10) quantities and cross zero); prints a JSON
summary.  Test infrastructure (uses oracle/).
usage: python tools/soak.py [--songs 512] [--seed 1] [--max-seconds 40]"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

INTS = ("start", "end", "mean", "variance", "n_frames", "nb_frames", "n_windows", "beat", "calm_or_loud")
FLOATS = ("tempo", "amplitude", "frequency", "attack", "force")


def make_song(seed, max_seconds):
    rng = np.random.default_rng(seed)
    rate = int(rng.choice([8000, 11025, 22050, 44100, 48000]))
    ch = int(rng.integers(1, 3))
    secs = float(rng.uniform(1.0, max_seconds))
    frames = max(int(rate * secs), 5120 // ch + 1)
    n = frames * ch + int(rng.integers(0, 7))          # ragged tails
    t = np.arange(frames) / rate
    level = 10 ** rng.uniform(1.5, 4.4)                 # 30 .. 25 000 LSB
    x = np.zeros(frames)
    for _ in range(int(rng.integers(1, 5))):
        f = 10 ** rng.uniform(1.5, np.log10(rate / 2.2))
        x += rng.uniform(0.2, 1.0) * np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28))
    bpm = rng.uniform(50, 200)
    x *= 0.55 + 0.45 * (np.sin(2 * np.pi * bpm / 60 * t) > 0.6)
    x = x / max(np.abs(x).max(), 1e-9) * level
    x += rng.normal(0, level * 10 ** rng.uniform(-3, -0.5), frames)
    pcm = np.empty(frames * ch)
    for c in range(ch):
        pcm[c::ch] = x * rng.uniform(0.6, 1.0) + rng.normal(0, 2.0, frames)
    pcm = np.concatenate([pcm, rng.normal(0, level * 0.1, n - frames * ch)])
    pcm += rng.choice([0, 0, 0, rng.uniform(-16000, 16000)])     # occasional DC (mean / variance wrap paths)
    pcm = np.clip(np.rint(pcm), -32768, 32767).astype(np.int16)
    if rng.random() < 0.3:
        pcm[: int(rng.integers(1, 3000))] = 0
    if rng.random() < 0.3:
        pcm[-int(rng.integers(1, 3000)):] = 0
    if rng.random() < 0.2:
        a = int(rng.integers(0, n // 2))
        pcm[a:a + int(rng.integers(1, n // 4))] = 0
    if not pcm.any():
        pcm[n // 2] = 1
    return pcm, ch, max(1, int(secs))


_orc = None


def _oracle_one(args):
    global _orc
    if _orc is None:
        from oracle_py import Oracle
        _orc = Oracle()
    seed, max_seconds = args
    pcm, ch, dur = make_song(seed, max_seconds)
    return seed, _orc.analyze(pcm, ch, dur)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=512)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-seconds", type=float, default=40.0)
    ap.add_argument("--procs", type=int, default=0)
    a = ap.parse_args()
    import bliss_amd
    seeds = [a.seed * 1000003 + i for i in range(a.songs)]
    t0 = time.time()
    songs = [make_song(s, a.max_seconds) for s in seeds]
    corpus = bliss_amd.DeviceCorpus([p.size for p, _, _ in songs], [c for _, c, _ in songs],
                                    [d for _, _, d in songs])
    for i, (p, _, _) in enumerate(songs):
        corpus.upload(i, p)
    corpus.analyze()
    got = corpus.fetch()
    t1 = time.time()
    with mp.get_context("fork").Pool(a.procs or os.cpu_count()) as pool:
        refs = dict(pool.imap_unordered(_oracle_one, [(s, a.max_seconds) for s in seeds], chunksize=1))
    t2 = time.time()
    bad_int, bad_float, not_bitwise, worst = [], [], 0, 0.0
    worst_by = {k: 0.0 for k in FLOATS}
    nbit_by = {k: 0 for k in FLOATS}
    min_margin = 1e9
    for i, s in enumerate(seeds):
        r, g = refs[s], got[i]
        min_margin = min(min_margin, r["min_peak_margin"])
        for k in INTS:
            if int(g[k]) != int(r[k]):
                bad_int.append((s, k, int(g[k]), int(r[k])))
        for k in FLOATS:
            x, y = float(g[k]), float(r[k])
            worst = max(worst, abs(x - y))
            worst_by[k] = max(worst_by[k], abs(x - y))
            # frequency / force: the reference's own absolute 1e-5 on top of the relative bound (they
            # cross zero and go through an f32 DFT that is not the oracle's); the rest: strict relative
            tol = 1e-5 + 1e-4 * abs(y) if k in ("frequency", "force") else 1e-4 * max(abs(y), 1e-6)
            if abs(x - y) > tol:
                bad_float.append((s, k, x, y))
            if np.float32(x) != np.float32(y):
                not_bitwise += 1
                nbit_by[k] += 1
    print(json.dumps({"songs": a.songs, "seed": a.seed, "int_mismatches": bad_int[:10],
                      "n_int_mismatches": len(bad_int), "float_out_of_tolerance": bad_float[:10],
                      "n_float_out_of_tolerance": len(bad_float), "float_fields_not_bit_identical": not_bitwise,
                      "worst_abs_err": worst, "worst_abs_err_by_field": worst_by, "not_bit_identical_by_field": nbit_by, "min_peak_margin": min_margin,
                      "gpu_seconds_incl_synthesis_and_upload": round(t1 - t0, 2),
                      "oracle_seconds": round(t2 - t1, 2), "procs": a.procs or os.cpu_count(),
                      "tolerance": "ints exact; tempo/amplitude/attack 1e-4 rel; frequency/force 1e-5 + 1e-4 |ref|"}))
    return 1 if (bad_int or bad_float) else 0


if __name__ == "__main__":
    sys.exit(main())
