#!/usr/bin/env python3
"""Throughput of the host-buffer entry point (bl_amd_analyze_batch_host): decoded PCM in ordinary host
memory -> pinned staging -> hipMemcpyAsync overlapped with the kernels -> results on the host.
This is the PCIe-inclusive rate of DESIGN.md section 5; it is never bench.py's `value`.
usage: python tools/host_path_bench.py [--songs 256] [--seconds 180]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=256)
    ap.add_argument("--seconds", type=int, default=180)
    ap.add_argument("--distinct", type=int, default=8, help="distinct PCM buffers cycled through the batch")
    a = ap.parse_args()
    import bliss_amd
    rng = np.random.default_rng(3)
    n = 44100 * 2 * a.seconds
    t = np.arange(n // 2) / 44100.0
    bufs = []
    for i in range(a.distinct):
        x = 6000 * np.sin(2 * np.pi * (200 + 37 * i) * t) * (0.6 + 0.4 * (np.sin(2 * np.pi * 2 * t) > 0))
        x = np.repeat(x, 2) + rng.normal(0, 300, n)
        bufs.append(np.clip(np.rint(x), -32768, 32767).astype(np.int16))
    pcm = [bufs[i % a.distinct] for i in range(a.songs)]
    bliss_amd.analyze_batch_host(pcm[: min(160, a.songs)], 2, a.seconds)         # warm-up: both pinned buffers at full size
    t0 = time.perf_counter()
    res = bliss_amd.analyze_batch_host(pcm, 2, a.seconds)
    dt = time.perf_counter() - t0
    ok = bool(np.all(res["status"] == 0))
    same = all(res["tempo"][i] == res["tempo"][i % a.distinct] and res["attack"][i] == res["attack"][i % a.distinct]
               for i in range(a.songs))
    print(json.dumps({"songs": a.songs, "seconds_per_song": a.seconds, "wall_s": round(dt, 3),
                      "songs_per_s": round(a.songs / dt, 1), "GB_per_s_pcm": round(a.songs * n * 2 / dt / 1e9, 2),
                      "status_ok": ok, "repeats_identical": same}))


if __name__ == "__main__":
    main()
