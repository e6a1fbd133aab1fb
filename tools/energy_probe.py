#!/usr/bin/env python3
"""Socket power and shader clock per kind of work: every stream of the issue micro-benchmark (tools/gen_ubench_issue.py)
runs alone on all 256 CUs for a few seconds (`ubench_issue.bin <name> long <waves per SIMD> <seconds>`) while
bench.DeviceState samples the device; printed per stream: wave-instructions per second of the whole chip, mean clock
and power in the settled part of the run, and the energy per wave-instruction above the clocked-but-idle baseline
(the `idle_nop` stream).  What the envelope window kernel is limited by is the 1.4 kW cap: this
says what each class of its instructions costs against it.
usage: python tools/energy_probe.py [--seconds 4] [--waves 2] [--streams fma_f64_svv,add_f64,...]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEFAULT = ("idle_nop,fma_f64_vvv,fma_f64_svv,add_f64,mul_f64,cvt_f64_i32,cvt_f32_f64,mov_dpp,sub_sdwa,add_u32,mov_b32,"
           "lds_read_b128,lds_write_b128,fma_with_lds_xchg,fir_x4,mix_3f64_1dpp")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--waves", type=int, default=2)
    ap.add_argument("--streams", default=DEFAULT)
    a = ap.parse_args()
    src, exe = "/tmp/ubench_issue.hip", "/tmp/ubench_issue.bin"
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_ubench_issue.py")], stdout=subprocess.PIPE,
                         text=True, check=True).stdout
    open(src, "w").write(gen)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-w", src, "-o", exe], check=True)
    from bench import DeviceState
    smp = DeviceState(DeviceState.pci_address(0), period=0.01)
    smp.start()
    out = {"seconds": a.seconds, "waves_per_simd": a.waves, "streams": {}}
    for name in [s for s in a.streams.split(",") if s]:
        time.sleep(1.0)
        t0 = time.perf_counter()
        r = subprocess.run([exe, name, "long", str(a.waves), str(a.seconds)], stdout=subprocess.PIPE, text=True)
        t1 = time.perf_counter()
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            out["streams"][name] = {"error": r.stdout[-200:]}
            continue
        e = json.loads(line[-1])
        dt = t1 - t0
        st = smp.summary(t0 + 0.45 * dt, t1 - 0.1 * dt)   # the process start and the first launches are not the stream
        e.update(sclk_mhz=(st.get("sclk_mhz") or {}).get("mean"), power_w=(st.get("power_w") or {}).get("mean"),
                 power_w_max=(st.get("power_w") or {}).get("max"), samples=st["samples"])
        if e.get("sclk_mhz"):
            e["cycles_per_wave_instr_per_simd"] = 1024 * e["sclk_mhz"] * 1e6 / e["wave_instr_per_s"]
        out["streams"][name] = e
    smp.stop_flag = True
    base = out["streams"].get("idle_nop", {}).get("power_w")
    if base:
        for name, e in out["streams"].items():
            if name != "idle_nop" and e.get("power_w"):
                e["nJ_per_wave_instr_above_idle"] = (e["power_w"] - base) / e["wave_instr_per_s"] * 1e9
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
