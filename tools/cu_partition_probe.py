#!/usr/bin/env python3
"""Experiment (round 6): does running the frequency pass of one half of a batch BESIDE the window kernel of the other
half, on disjoint sets of CUs, beat running them one after the other on all CUs?  The window kernel runs power-capped
 and fills a CU's LDS, so nothing can share a CU with it; what a partition could buy is clock:
a mix of f64-heavy and f32 / LDS-heavy CUs may draw less than 256 f64-heavy ones.

Two explicit contexts, each with its own resident half of the corpus and its own stream:
  plain      one context, the whole corpus, one stream (the product's way)
  two        two contexts on two ordinary streams, started together
  two_stag   the same, the second started one frequency pass later
  masked     two contexts on two streams created with hipExtStreamCreateWithCUMask (complementary halves of the CU
             mask), started together / staggered
Prints one JSON object: ms per whole corpus for each variant (median of --reps), shader clock and power.
usage: python tools/cu_partition_probe.py [--songs 1024] [--seconds 180] [--reps 7]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=1024)
    ap.add_argument("--seconds", type=int, default=180)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--loops", type=int, default=6, help="analyses per timed repetition")
    a = ap.parse_args()
    import torch
    import bliss_amd
    import bench
    lib = bliss_amd.load()
    assert lib.bl_amd_init(0) == 0
    hip = C.CDLL("libamdhip64.so")
    n = 44100 * 2 * a.seconds
    half = a.songs // 2
    whole = bliss_amd.DeviceCorpus([n] * a.songs, 2, a.seconds)
    whole.synth(seed_base=100000, sample_rate=44100)
    parts = []
    for h in range(2):
        c = bliss_amd.DeviceCorpus([n] * half, 2, a.seconds)
        c.synth(seed_base=100000 + h * half, sample_rate=44100)
        parts.append(c)
    ctxs = [bliss_amd.Context(0), bliss_amd.Context(0)]
    torch.cuda.synchronize()

    def masked_stream(bits):
        words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
        s = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
        assert rc == 0, rc
        return torch.cuda.ExternalStream(s.value)

    plain = [torch.cuda.Stream(), torch.cuda.Stream()]
    lo = (1 << 128) - 1
    masks = {"masked_128_128": (lo, lo << 128),
             "masked_interleaved": (int("01" * 128, 2), int("10" * 128, 2))}
    mstreams = {k: (masked_stream(v[0]), masked_stream(v[1])) for k, v in masks.items()}

    # how long one frequency pass of half the corpus takes (for the stagger): from the library's own event timing
    lib.bl_amd_profile_reset(); lib.bl_amd_profile(1)
    whole.analyze(); torch.cuda.synchronize()
    lib.bl_amd_profile(0)
    k = C.c_int(0)
    f_ms = 0.5 * lib.bl_amd_profile_ms(b"freq_scan", C.byref(k)) / max(k.value, 1)   # half the corpus on all CUs

    def run_plain():
        for _ in range(a.loops):
            whole.analyze()

    def run_two(streams, stagger):
        if stagger:
            with torch.cuda.stream(streams[1]):
                torch.cuda._sleep(int(f_ms * 1e-3 * 1e8))   # the sleep counts a 100 MHz clock
        for _ in range(a.loops):
            for h in range(2):
                with torch.cuda.stream(streams[h]):
                    parts[h].analyze(ctxs[h])

    def run_two_delay(streams, delay_ms):
        with torch.cuda.stream(streams[1]):
            torch.cuda._sleep(int(delay_ms * 1e-3 * 1e8))   # the sleep counts a 100 MHz clock
        for _ in range(a.loops):
            for h in range(2):
                with torch.cuda.stream(streams[h]):
                    parts[h].analyze(ctxs[h])

    variants = {"plain": run_plain, "two": lambda: run_two(plain, False), "two_stag": lambda: run_two(plain, True)}
    for k_, st in mstreams.items():
        variants[k_] = (lambda st=st: run_two(st, False))
        variants[k_ + "_stag"] = (lambda st=st: run_two(st, True))
    def run_whole_on(stream):
        with torch.cuda.stream(stream):
            for _ in range(a.loops):
                whole.analyze()
    variants["whole_on_one_128cu_mask"] = lambda: run_whole_on(mstreams["masked_128_128"][0])   # is the mask honoured?
    for frac in (1.0, 2.0):   # stagger by one / two frequency passes of a half on half the CUs
        variants[f"masked_128_128_stag_x{frac:g}"] = (lambda frac=frac: (torch.cuda.synchronize(), run_two_delay(mstreams["masked_128_128"], 2 * frac * f_ms)))
    out = {"songs": a.songs, "seconds": a.seconds, "loops": a.loops, "freq_scan_ms_half": f_ms, "variants": {}}
    ref = whole.fetch() if False else None
    for name, fn in variants.items():
        fn(); torch.cuda.synchronize()
        ts = []
        ds = bench.DeviceState(bench.DeviceState.pci_address(0), period=0.005)
        ds.start()
        t_begin = time.perf_counter()
        for _ in range(a.reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0) / a.loops)
        t_end = time.perf_counter()
        ds.stop_flag = True
        st = ds.summary(t_begin, t_end)
        out["variants"][name] = {"ms_per_corpus_median": float(np.median(ts)), "ms_all": [round(x, 3) for x in ts],
                                 "sclk_mhz": (st.get("sclk_mhz") or {}).get("mean"), "power_w": (st.get("power_w") or {}).get("mean")}
    # the halves analysed on the masked streams give the records of the whole
    gw = whole.fetch()
    g = np.concatenate([parts[0].fetch(), parts[1].fetch()])
    out["records_equal_to_plain"] = bool(all(np.array_equal(gw[f], g[f]) for f in gw.dtype.names if f != "atk_sum"))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
