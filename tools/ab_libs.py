#!/usr/bin/env python3
"""A/B of whole builds of libbliss_amd.so on one box: every library given is loaded in its own process (BLISS_AMD_LIB),
analyses the same resident corpus and prints its per-kernel HIP-event times; the libraries take turns --rounds times
over (boxes drift by a per cent or two within a run, medians do not), and the records of every library are compared
field by field with the first one's.  Prints one JSON object.
usage: python tools/ab_libs.py --libs bliss_amd/libbliss_amd.so,gpurun_out/old/libbliss_amd.so [--songs 1024]
       [--seconds 180] [--reps 5] [--rounds 3] [--fir-mode 2]"""
import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ["env_windows", "freq_scan", "tail", "amp", "scan", "freq"]


def child(a):
    sys.path.insert(0, ROOT)
    import torch
    import bliss_amd
    lib = bliss_amd.load()
    if a.fir_mode >= 0:
        lib.bl_amd_set_fir_mode(a.fir_mode)
    n = 44100 * 2 * a.seconds
    corpus = bliss_amd.DeviceCorpus([n] * a.songs, 2, a.seconds)
    corpus.synth(seed_base=100000, sample_rate=44100)
    torch.cuda.synchronize()
    corpus.analyze()
    got = corpus.fetch()
    lib.bl_amd_profile_reset()
    lib.bl_amd_profile(1)
    import time
    t0 = time.perf_counter()
    for _ in range(a.reps):
        corpus.analyze()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / a.reps
    lib.bl_amd_profile(0)
    ms = {}
    k = C.c_int(0)
    for nm in KERNELS:
        v = lib.bl_amd_profile_ms(nm.encode(), C.byref(k))
        if k.value:
            ms[nm] = v / k.value
    h = hashlib.sha256()
    for f in got.dtype.names:  # field by field: the records carry padding nothing writes
        h.update(np.ascontiguousarray(got[f]).tobytes())
    print(json.dumps({"ms": ms, "wall_ms": wall, "records_sha256": h.hexdigest(),
                      "status_max": int(got["status"].max())}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--songs", type=int, default=1024)
    ap.add_argument("--seconds", type=int, default=180)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--fir-mode", type=int, default=-1)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a)
    libs = [os.path.abspath(p) for p in a.libs.split(",") if p]
    res = {p: [] for p in libs}
    for _ in range(a.rounds):
        for p in libs:
            env = dict(os.environ, BLISS_AMD_LIB=p)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--songs", str(a.songs), "--seconds",
                                  str(a.seconds), "--reps", str(a.reps), "--fir-mode", str(a.fir_mode)],
                                 env=env, capture_output=True, text=True, check=True).stdout
            res[p].append(json.loads(out.strip().splitlines()[-1]))
    base = res[libs[0]][0]["records_sha256"]
    rep = {"songs": a.songs, "seconds": a.seconds, "reps": a.reps, "rounds": a.rounds, "fir_mode": a.fir_mode, "libs": {}}
    for p in libs:
        r = res[p]
        rep["libs"][os.path.relpath(p, ROOT)] = {
            "ms_median": {k: round(float(np.median([x["ms"][k] for x in r if k in x["ms"]])), 3) for k in KERNELS
                          if any(k in x["ms"] for x in r)},
            "env_windows_ms_all": [round(x["ms"].get("env_windows", 0.0), 3) for x in r],
            "wall_ms_median": round(float(np.median([x["wall_ms"] for x in r])), 3),
            "records_identical_to_first": all(x["records_sha256"] == base for x in r),
            "status_max": max(x["status_max"] for x in r)}
    b = rep["libs"][os.path.relpath(libs[0], ROOT)]["ms_median"]
    for p in libs[1:]:
        e = rep["libs"][os.path.relpath(p, ROOT)]
        e["vs_first"] = {k: round(e["ms_median"][k] / b[k], 4) for k in e["ms_median"] if k in b and b[k] > 0}
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
