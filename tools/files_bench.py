#!/usr/bin/env python3
"""The reference's corpus loop on files (`for f: bl_analyze(f)`) against bl_amd_analyze_files:
N WAV files (44.1 kHz s16 stereo, --seconds each) written to a scratch directory, analysed one by
one through bl_analyze and in one call with 1 / 4 / 16 / 0 (= all) decoder threads.  The work per
file is the host's — file read, rate conversion to 22 050 Hz — so this is a host-bound number; it
shows what overlapping decode, transfer and analysis buys over the sequential loop.
Note: a 44.1 kHz s16 file goes through the converter's 16-bit path — parity unpinned (s16 path).
usage: python tools/files_bench.py [--files 256] [--seconds 60]"""
import argparse
import ctypes as C
import json
import os
import struct
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=256)
    ap.add_argument("--seconds", type=int, default=60)
    ap.add_argument("--dir", default="")
    a = ap.parse_args()
    import bliss_amd
    from bliss_amd import _lib
    from tests.oracle_py import Oracle
    lib = bliss_amd.load()
    orc = Oracle()
    d = a.dir or tempfile.mkdtemp(prefix="bl_files_")
    rate, ch = 44100, 2
    names = []
    for i in range(a.files):
        pcm = orc.synth(70000 + i, rate, ch, rate * ch * a.seconds)   # integer generator of the bench corpus
        raw = pcm.tobytes()
        fmt = struct.pack("<HHIIHH", 1, ch, rate, rate * ch * 2, ch * 2, 16)
        body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(raw)) + raw
        f = os.path.join(d, f"s{i:05d}.wav")
        open(f, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)
        names.append(f)
    out = {"files": a.files, "seconds_per_file": a.seconds, "file_MB": round(os.path.getsize(names[0]) / 1e6, 1),
           "parity": "parity unpinned (s16 path): 44.1 kHz s16 files pass through the restated 16-bit converter"}
    song = _lib.BlSong()
    lib.bl_analyze(names[0].encode(), C.byref(song)); lib.bl_free_song(C.byref(song))   # warm-up
    t0 = time.perf_counter()
    seq = []
    for f in names:
        lib.bl_analyze(f.encode(), C.byref(song))
        seq.append((song.force_vector.tempo, song.force_vector.amplitude, song.force_vector.frequency, song.force_vector.attack))
        lib.bl_free_song(C.byref(song))
    dt = time.perf_counter() - t0
    out["bl_analyze_loop"] = {"wall_s": round(dt, 3), "files_per_s": round(a.files / dt, 1)}
    for thr in (1, 4, 16, 0):
        t0 = time.perf_counter()
        recs, codes = bliss_amd.analyze_files(names, n_threads=thr)
        dt = time.perf_counter() - t0
        same = all(r is not None and tuple(r["force_vector"][k] for k in ("tempo", "amplitude", "frequency", "attack")) == s
                   for r, s in zip(recs, seq))
        out[f"analyze_files_threads_{thr or 'all'}"] = {"wall_s": round(dt, 3), "files_per_s": round(a.files / dt, 1),
                                                         "identical_to_loop": same}
    for f in names:
        os.remove(f)
    if not a.dir:
        os.rmdir(d)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
