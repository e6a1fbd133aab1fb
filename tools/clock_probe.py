#!/usr/bin/env python3
"""What clock and power the kernels really run at.  A resident corpus is analysed in a loop while bench.DeviceState
samples the amdgpu hwmon files of the device (shader clock, socket power, temperature); printed: their distribution
during the analysis (80 % of it is the envelope window kernel) next to the kernels' HIP-event times, and the same for
the integer synthesis kernel for comparison.  First measured in round 5: the window kernel does not run at the 2.4 GHz
the part is specified with — 2.16-2.23 GHz at 1.15-1.35 kW (cap 1.4 kW) on the boxes met, the synthesis kernel 2.38 GHz
at 0.87 kW — which is what the spread of its time over the boxes of the pool (268-292 ms per 8 192 songs) follows.
usage: python tools/clock_probe.py [--songs 1024] [--seconds 180] [--loops 40]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=1024)
    ap.add_argument("--seconds", type=int, default=180)
    ap.add_argument("--loops", type=int, default=40)
    a = ap.parse_args()
    import torch
    import bliss_amd
    from bench import DeviceState
    lib = bliss_amd.load()
    n = 44100 * 2 * a.seconds
    corpus = bliss_amd.DeviceCorpus([n] * a.songs, 2, a.seconds)
    corpus.synth(seed_base=100000, sample_rate=44100)
    torch.cuda.synchronize()
    corpus.analyze()
    torch.cuda.synchronize()
    smp = DeviceState(DeviceState.pci_address(0), period=0.002)
    smp.start()
    out = {"songs": a.songs, "seconds": a.seconds, "loops": a.loops}

    def timed(name, fn):
        time.sleep(0.3)
        lib.bl_amd_profile_reset()
        lib.bl_amd_profile(1)
        t0 = time.perf_counter()
        for _ in range(a.loops):
            fn()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        lib.bl_amd_profile(0)
        k = C.c_int(0)
        ms = {}
        for nm in (b"env_windows", b"freq_scan", b"env_tail", b"amp_finish"):
            v = lib.bl_amd_profile_ms(nm, C.byref(k))
            if k.value:
                ms[nm.decode()] = round(v / k.value, 3)
        out[name] = dict(smp.summary(t0 + 0.05, t1), wall_ms_per_loop=round((t1 - t0) * 1e3 / a.loops, 3), kernel_ms=ms)

    timed("analyze", corpus.analyze)
    timed("synth", lambda: corpus.synth(seed_base=100000, sample_rate=44100))
    timed("analyze_again", corpus.analyze)
    smp.stop_flag = True
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
