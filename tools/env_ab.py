#!/usr/bin/env python3
"""A/B of the phase-priority tables of k_env_windows3 (measurement build, bl_amd_measure_env): the same resident
corpus analysed with every table, results compared field by field with the shipped table's, the kernel timed with
HIP events, and — for --probe tables — the s_memtime stamps of workgroup (0, 0) summarised per wave: where a
compute wave's round goes (arithmetic phases, exchange phases, the wait in front of the hand-over).
A table is six hex digits, one priority (0..3) per phase, phase 0 in the lowest digit; the measurement build
instantiates the ones of EV_PRIO_TABS (bl_kernels.hip) beside the shipped 222011.  Prints one JSON object.
usage: python tools/env_ab.py [--songs 1024] [--seconds 180] [--tabs 000000,111111,322110] [--probe 222011] [--reps 3]"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBE_ROUNDS, PROBE_SLOTS = 16, 12
SLOTS = ["start", "fir_done", "inputs_loaded", "fft1_done", "exchanged", "fft2_done", "rows_free", "published"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=1024)
    ap.add_argument("--seconds", type=int, default=180)
    ap.add_argument("--tabs", default="000000,111111,322110,321000,222110,222111,232011,222112")
    ap.add_argument("--probe", default="")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--dump", default="", help="directory for the raw stamps (probe<v>.npy)")
    a = ap.parse_args()
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "bliss_amd", "csrc"), "measure"], check=True)
    os.environ["BLISS_AMD_LIB"] = os.path.join(ROOT, "bliss_amd", "libbliss_amd_measure.so")
    import torch
    import bliss_amd
    lib = bliss_amd.load()
    lib.bl_amd_measure_env.argtypes = [C.c_int, C.c_void_p]
    n = 44100 * 2 * a.seconds
    corpus = bliss_amd.DeviceCorpus([n] * a.songs, 2, a.seconds)
    corpus.synth(seed_base=100000, sample_rate=44100)
    torch.cuda.synchronize()
    probe = torch.zeros(8 * PROBE_ROUNDS * PROBE_SLOTS, dtype=torch.int64, device="cuda")

    def run(tab, with_probe=False):
        probe.zero_()
        var = -1 if tab is None else (int(tab, 16) | ((1 << 24) if with_probe else 0))
        assert lib.bl_amd_measure_env(var, C.c_void_p(probe.data_ptr()) if with_probe else None) == 0
        corpus.analyze()
        got = corpus.fetch()
        assert int(got["status"].max()) == 0, f"table {tab} is not instantiated (EV_PRIO_TABS)"
        lib.bl_amd_profile_reset()
        lib.bl_amd_profile(1)
        for _ in range(a.reps):
            corpus.analyze()
        torch.cuda.synchronize()
        lib.bl_amd_profile(0)
        k = C.c_int(0)
        ms = lib.bl_amd_profile_ms(b"env_windows", C.byref(k))
        return got, ms / max(k.value, 1)

    base, base_ms = run(None)

    def diff(got):  # field by field, bit patterns (the records carry 4 bytes of padding that nothing writes)
        bits = lambda x: x.view(np.int32) if x.dtype == np.float32 else x.view(np.int64) if x.dtype == np.float64 else x
        return {k: int(np.count_nonzero(bits(got[k]) != bits(base[k]))) for k in got.dtype.names
                if np.count_nonzero(bits(got[k]) != bits(base[k]))}
    # the tables take turns, --rounds times over: boxes drift by a per cent or two within a run, medians do not
    tabs = [t for t in a.tabs.split(",") if t]
    times = {t: [] for t in ["shipped"] + tabs}
    same = {}
    times["shipped"].append(base_ms)
    for r in range(a.rounds):
        for tab in tabs:
            got, ms = run(tab)
            times[tab].append(ms)
            same[tab] = same.get(tab, True) and not diff(got)
        again, ms = run(None)
        times["shipped"].append(ms)
        same["shipped"] = same.get("shipped", True) and not diff(again)
    med = {t: float(np.median(v)) for t, v in times.items()}
    out = {"songs": a.songs, "seconds": a.seconds, "rounds": a.rounds,
           "tables": {t: {"env_windows_ms_median": med[t], "env_windows_ms_all": [round(x, 3) for x in times[t]],
                          "vs_shipped": med[t] / med["shipped"], "records_identical": same[t]} for t in times}}
    for tab in [t for t in a.probe.split(",") if t]:
        got, ms = run(tab, True)
        st = probe.cpu().numpy().reshape(8, PROBE_ROUNDS, PROBE_SLOTS)
        if a.dump:
            np.save(os.path.join(a.dump, f"probe_{tab}.npy"), st)
        rep = {"env_windows_ms": ms, "records_identical": not diff(got), "waves": {}}
        for w in range(7):
            t = st[w, 4:PROBE_ROUNDS, :8].astype(np.float64)  # steady-state rounds
            if not t[:, 0].all():
                continue
            d = np.diff(t, axis=1).mean(axis=0)
            period = np.diff(t[:, 0]).mean()
            rep["waves"][str(w)] = dict({f"{SLOTS[i]}->{SLOTS[i + 1]}": round(float(d[i])) for i in range(7)},
                                        round_period=round(float(period)),
                                        start_offset_vs_wave0=round(float((t[:, 0] - st[0, 4:PROBE_ROUNDS, 0]).mean())))
        t = st[7, 4:PROBE_ROUNDS, :3].astype(np.float64)
        if t[:, 0].all():
            rep["summing_wave"] = {"wait_for_tile": round(float((t[:, 1] - t[:, 0]).mean())),
                                   "pass": round(float((t[:, 2] - t[:, 1]).mean())),
                                   "period": round(float(np.diff(t[:, 0]).mean()))}
        out["tables"][f"probe:{tab}"] = rep
    lib.bl_amd_measure_env(-1, None)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
