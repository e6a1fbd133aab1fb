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
    ap.add_argument("--mode", choices=("staged", "registered"), default="staged")
    ap.add_argument("--latency", action="store_true", help="also: raw pinned H2D rate, single-song latencies")
    ap.add_argument("--native-rate", type=int, default=0,
                    help="treat the buffers as PCM at this rate: bl_amd_analyze_batch_host_rate converts each "
                         "wave to 22 050 Hz on the device before the analysis")
    a = ap.parse_args()
    import bliss_amd
    lib = bliss_amd.load()
    assert lib.bl_amd_set_host_transfer(1 if a.mode == "registered" else 0) == 0
    rng = np.random.default_rng(3)
    n = 44100 * 2 * a.seconds
    t = np.arange(n // 2) / 44100.0
    bufs = []
    for i in range(a.distinct):
        x = 6000 * np.sin(2 * np.pi * (200 + 37 * i) * t) * (0.6 + 0.4 * (np.sin(2 * np.pi * 2 * t) > 0))
        x = np.repeat(x, 2) + rng.normal(0, 300, n)
        bufs.append(np.clip(np.rint(x), -32768, 32767).astype(np.int16))
    if a.mode == "registered":  # in-place pinning needs one buffer per song (a range is registered once)
        bufs = bufs + [bufs[i % a.distinct].copy() for i in range(a.distinct, a.songs)]
        a.distinct_eff = a.distinct
        pcm = bufs[: a.songs]
    else:
        pcm = [bufs[i % a.distinct] for i in range(a.songs)]
    if a.native_rate:
        run = lambda songs: bliss_amd.analyze_batch_host_rate(songs, 2, a.seconds, a.native_rate)
    else:
        run = lambda songs: bliss_amd.analyze_batch_host(songs, 2, a.seconds)
    run(pcm[: min(160, a.songs)])         # warm-up: both pinned buffers at full size
    t0 = time.perf_counter()
    res = run(pcm)
    dt = time.perf_counter() - t0
    ok = bool(np.all(res["status"] == 0))
    same = all(res["tempo"][i] == res["tempo"][i % a.distinct] and res["attack"][i] == res["attack"][i % a.distinct]
               for i in range(a.songs))
    line = {"mode": a.mode, "songs": a.songs, "seconds_per_song": a.seconds, "wall_s": round(dt, 3),
            "songs_per_s": round(a.songs / dt, 1), "GB_per_s_pcm": round(a.songs * n * 2 / dt / 1e9, 2),
            "status_ok": ok, "repeats_identical": same}
    if a.native_rate:
        line["native_rate"] = a.native_rate
    if a.latency:
        import ctypes as C
        import torch
        from bliss_amd import _lib
        # the link itself: one pinned 2 GB buffer, host -> device
        h = torch.empty(1 << 30, dtype=torch.int16).pin_memory()
        d = torch.empty_like(h, device="cuda")
        d.copy_(h, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        line["raw_pinned_h2d_GB_per_s"] = round(5 * h.numel() * 2 / (time.perf_counter() - t0) / 1e9, 2)
        del h, d
        # one song at a time: the drop-in bl_analyze() on the reference's fixture, and one S180 buffer
        flac = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                            "song.flac").encode()
        song = _lib.BlSong()
        lib.bl_analyze(flac, C.byref(song)); lib.bl_free_song(C.byref(song))
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            lib.bl_analyze(flac, C.byref(song))
            ts.append(time.perf_counter() - t0)
            lib.bl_free_song(C.byref(song))
        line["bl_analyze_song_flac_ms"] = round(1e3 * min(ts), 2)
        lib.bl_audio_decode(flac, C.byref(song))
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            lib.bl_audio_decode(flac, C.byref(song))
            ts.append(time.perf_counter() - t0)
            lib.bl_free_song(C.byref(song))
        line["bl_audio_decode_song_flac_ms"] = round(1e3 * min(ts), 2)
        bliss_amd.analyze_batch_host(pcm[:1], 2, a.seconds)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            bliss_amd.analyze_batch_host(pcm[:1], 2, a.seconds)
            ts.append(time.perf_counter() - t0)
        line["one_song_batch_host_ms"] = round(1e3 * min(ts), 2)
        va, vb = _lib.ForceVector(1, 2, 3, 4), _lib.ForceVector(4, 3, 2, 1.5)
        t0 = time.perf_counter()
        for _ in range(100000):
            lib.bl_distance(va, vb)
        line["bl_distance_call_us_incl_ctypes"] = round(1e6 * (time.perf_counter() - t0) / 100000, 3)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
