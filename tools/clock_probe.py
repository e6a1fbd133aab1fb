#!/usr/bin/env python3
"""What clock and power the envelope window kernel really runs at.  A resident corpus is analysed in a loop while a
thread samples the amdgpu hwmon / sysfs files of the device (shader clock, socket power, temperature); printed: the
distribution of the samples taken while the window kernel was running, next to the kernel's HIP-event time — and
the same for the statistics / frequency pass alone (an HBM-bound kernel) for comparison.
usage: python tools/clock_probe.py [--songs 1024] [--seconds 180] [--loops 40] [--lib measure|product]"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def find_files(bdf=None):
    """hwmon files of the device with PCI address `bdf` (as hipDeviceGetPCIBusId prints it); without one, of every
    amdgpu card that has them (the caller keeps the busiest)"""
    cards = []
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        dev = os.path.join(card, "device")
        hws = glob.glob(os.path.join(dev, "hwmon", "hwmon*"))
        if not hws or not os.path.exists(os.path.join(hws[0], "freq1_input")):
            continue
        out = {"card": card, "bdf": os.path.basename(os.path.realpath(dev))}
        for f in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input", "temp2_input", "power1_cap"):
            p = os.path.join(hws[0], f)
            if os.path.exists(p):
                out[f] = p
        for f in ("pp_dpm_sclk", "pp_dpm_mclk", "power_dpm_force_performance_level"):
            p = os.path.join(dev, f)
            if os.path.exists(p):
                out[f] = p
        cards.append(out)
    if bdf:
        hit = [c for c in cards if c["bdf"].lower() == bdf.lower()]
        if hit:
            return hit
    return cards


def rd(p):
    try:
        with open(p) as f:
            return f.read().strip()
    except OSError:
        return None


def cur_level(txt):
    """'0: 132Mhz\\n1: 2400Mhz *' -> 2400"""
    if not txt:
        return None
    for line in txt.splitlines():
        if line.rstrip().endswith("*"):
            try:
                return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
            except (IndexError, ValueError):
                return None
    return None


class Sampler(threading.Thread):
    def __init__(self, cards, period=0.002):
        super().__init__(daemon=True)
        self.cards, self.period, self.rows, self.stop = cards, period, [[] for _ in cards], False

    def run(self):
        while not self.stop:
            t = time.perf_counter()
            for f, rows in zip(self.cards, self.rows):
                row = {"t": t}
                v = rd(f["freq1_input"])
                row["sclk_mhz"] = float(v) / 1e6 if v else None
                for k in ("power1_average", "power1_input"):
                    if k in f:
                        v = rd(f[k])
                        row["power_w"] = float(v) / 1e6 if v else None
                        break
                for k in ("temp2_input", "temp1_input"):
                    if k in f:
                        v = rd(f[k])
                        row["temp_c"] = float(v) / 1e3 if v else None
                        break
                rows.append(row)
            dt = self.period - (time.perf_counter() - t)
            if dt > 0:
                time.sleep(dt)


def summarise(rows, t0, t1):
    import numpy as np
    sel = [r for r in rows if t0 <= r["t"] <= t1]
    out = {"samples": len(sel)}
    for k in ("sclk_mhz", "power_w", "temp_c"):
        v = np.array([r[k] for r in sel if r.get(k) is not None], dtype=float)
        if len(v):
            out[k] = {"mean": round(float(v.mean()), 1), "min": round(float(v.min()), 1), "p10": round(float(np.percentile(v, 10)), 1),
                      "p50": round(float(np.median(v)), 1), "p90": round(float(np.percentile(v, 90)), 1), "max": round(float(v.max()), 1)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=1024)
    ap.add_argument("--seconds", type=int, default=180)
    ap.add_argument("--loops", type=int, default=40)
    ap.add_argument("--lib", default="product")
    a = ap.parse_args()
    if a.lib == "measure":
        os.environ["BLISS_AMD_LIB"] = os.path.join(ROOT, "bliss_amd", "libbliss_amd_measure.so")
    import torch
    import bliss_amd
    lib = bliss_amd.load()
    bdf = None
    try:
        buf = C.create_string_buffer(64)
        hip = C.CDLL("libamdhip64.so")
        if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
            bdf = buf.value.decode()
    except OSError:
        pass
    cards = find_files(bdf)
    out = {"pci": bdf, "cards_sampled": [c["card"] for c in cards], "static": {}}
    n = 44100 * 2 * a.seconds
    corpus = bliss_amd.DeviceCorpus([n] * a.songs, 2, a.seconds)
    corpus.synth(seed_base=100000, sample_rate=44100)
    torch.cuda.synchronize()
    corpus.analyze()
    torch.cuda.synchronize()
    smp = Sampler(cards, 0.002 if len(cards) == 1 else 0.01)
    smp.start()
    time.sleep(0.3)
    t_idle = time.perf_counter()

    def busiest(t0, t1):  # the card that drew the most power in [t0, t1]
        best, best_p = 0, -1.0
        for i, rows in enumerate(smp.rows):
            s_ = summarise(rows, t0, t1).get("power_w", {}).get("mean", 0.0)
            if s_ > best_p:
                best, best_p = i, s_
        return best
    def timed(name, fn, loops):
        lib.bl_amd_profile_reset()
        lib.bl_amd_profile(1)
        t0 = time.perf_counter()
        for _ in range(loops):
            fn()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        lib.bl_amd_profile(0)
        k = C.c_int(0)
        ms = {}
        for nm in (b"env_windows", b"freq_scan", b"tail", b"amp"):
            v = lib.bl_amd_profile_ms(nm, C.byref(k))
            if k.value:
                ms[nm.decode()] = round(v / k.value, 3)
        i = busiest(t0 + 0.05, t1)
        out[name] = dict(summarise(smp.rows[i], t0 + 0.05, t1), card=cards[i]["card"], wall_ms_per_loop=round((t1 - t0) * 1e3 / loops, 3),
                         kernel_ms=ms)
        time.sleep(0.3)

    # the whole analysis (80 % of it is the window kernel), then the integer synthesis kernel for comparison
    timed("analyze", corpus.analyze, a.loops)
    timed("synth", lambda: corpus.synth(seed_base=100000, sample_rate=44100), a.loops)
    timed("analyze_again", corpus.analyze, a.loops)
    smp.stop = True
    i = busiest(t_idle, time.perf_counter())
    out["idle_before"] = summarise(smp.rows[i], t_idle - 0.3, t_idle)
    for k in ("pp_dpm_sclk", "pp_dpm_mclk", "power1_cap", "power_dpm_force_performance_level"):
        if k in cards[i]:
            out["static"][k] = rd(cards[i][k])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
