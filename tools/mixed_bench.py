#!/usr/bin/env python3
"""BASELINE configs[4] shape as a throughput run: songs of log-uniform length in [10 s, 600 s] at 44.1 kHz,
half mono / half stereo, synthesised on the device, analysed in one batch call; compares the aggregate PCM
rate with the fixed-length configs[2] run.  (Not bench.py's line: the contract quotes the metric on
configs[2]; this documents what mixed lengths cost.)
usage: python tools/mixed_bench.py [--songs 8192] [--steps 3]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    import torch
    import bliss_amd
    rng = np.random.default_rng(5)
    secs = np.exp(rng.uniform(np.log(10.0), np.log(600.0), a.songs))
    ch = rng.integers(1, 3, a.songs)
    lengths = (np.floor(secs * 44100).astype(np.int64) * ch).tolist()
    durs = np.maximum(1, np.floor(secs)).astype(np.int64).tolist()
    corpus = bliss_amd.DeviceCorpus(lengths, ch.tolist(), durs)
    corpus.synth(seed_base=0, sample_rate=44100)
    torch.cuda.synchronize()
    corpus.analyze()
    torch.cuda.synchronize()
    lib = bliss_amd.load()
    lib.bl_amd_profile_reset(); lib.bl_amd_profile(1)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        corpus.analyze()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    lib.bl_amd_profile(0)
    import ctypes as C
    kern = {}
    for name in ("pcm_scan", "freq_scan", "amp_finish", "freq_frames", "env_windows", "env_tail"):
        n = C.c_int(0)
        ms = lib.bl_amd_profile_ms(name.encode(), C.byref(n))
        kern[name] = round(ms / a.steps, 2)   # per batch: a mixed batch launches the window and tail kernels twice
    res = corpus.fetch()
    gb = corpus.pcm_bytes / 1e9
    print(json.dumps({"songs": a.songs, "pcm_GB": round(gb, 1), "mean_seconds": round(float(secs.mean()), 1),
                      "ms_per_batch": round(dt * 1e3, 1), "songs_per_s": round(a.songs / dt, 1),
                      "pcm_GB_per_s": round(gb / dt, 1),
                      "equivalent_S180_songs_per_s": round(gb * 1e9 / 31752000 / dt, 1),
                      "kernels_ms_per_batch": kern,
                      # what the batch takes beyond the kernels of the main stream: the part of the serial envelope
                      # tail (side stream) that nothing hides, plus the small kernels and launch gaps
                      "ms_beyond_main_stream_kernels": round(dt * 1e3 - sum(kern[k] for k in ("pcm_scan", "freq_scan", "amp_finish", "freq_frames", "env_windows")), 1),
                      "status_ok": bool(np.all(res["status"] == 0))}))


if __name__ == "__main__":
    main()
