#!/usr/bin/env python3
"""A/B of the envelope kernel's FIR modes on one corpus (BL_AMD_FIR_FUSED = 0 | 1 | 2):
mode 0 is the reference's unfused arithmetic (bit-identical to the CPU oracle in every parity
test), so comparing the other modes with it on the GPU measures what an f64-level change of the
FIR does downstream without any CPU time: how many f32 window energies move (and by how many
ulp), and whether any integer (`beat`) or float feature changes.  Also times the kernel per mode
(HIP events).  Prints one JSON object.
usage: python tools/fir_ab.py [--kind synth|soak] [--songs 2048] [--seconds 180] [--seed 1]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

INTS = ("start", "end", "mean", "variance", "n_frames", "nb_frames", "n_windows", "beat", "calm_or_loud", "status")
FLOATS = ("tempo", "amplitude", "frequency", "attack", "force")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="synth")
    ap.add_argument("--songs", type=int, default=2048)
    ap.add_argument("--seconds", type=int, default=180)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--modes", default="0,1,2")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch
    import bliss_amd
    lib = bliss_amd.load()
    if a.kind == "synth":
        n = 44100 * 2 * a.seconds
        corpus = bliss_amd.DeviceCorpus([n] * a.songs, 2, a.seconds)
        corpus.synth(seed_base=a.seed * 100000, sample_rate=44100)
    else:
        from soak import make_song
        seeds = [a.seed * 1000003 + i for i in range(a.songs)]
        songs = [make_song(s, float(a.seconds)) for s in seeds]
        corpus = bliss_amd.DeviceCorpus([p.size for p, _, _ in songs], [c for _, c, _ in songs],
                                        [d for _, _, d in songs])
        for i, (p, _, _) in enumerate(songs):
            corpus.upload(i, p)
    torch.cuda.synchronize()

    def run(mode):
        os.environ["BL_AMD_FIR_FUSED"] = str(mode)
        corpus.analyze()
        got = corpus.fetch()
        total = int(got["nb_frames"].astype(np.int64).sum())
        en = np.zeros(total, dtype=np.float32)
        assert lib.bl_amd_last_energies(en.ctypes.data_as(C.POINTER(C.c_float)), total) == total
        lib.bl_amd_profile_reset()
        lib.bl_amd_profile(1)
        for _ in range(a.reps):
            corpus.analyze()
        torch.cuda.synchronize()
        lib.bl_amd_profile(0)
        k = C.c_int(0)
        ms = lib.bl_amd_profile_ms(b"env_windows", C.byref(k))
        return got, en, ms / max(k.value, 1)

    modes = [int(m) for m in a.modes.split(",")]
    base_got, base_en, base_ms = run(0)
    n_windows = int(base_got["n_windows"].astype(np.int64).sum())
    out = {"kind": a.kind, "songs": a.songs, "seconds": a.seconds, "seed": a.seed,
           "windows": n_windows, "ordered_adds": n_windows * 257,
           "modes": {"0": {"env_windows_ms": base_ms}}}
    bi = base_en.view(np.int32).astype(np.int64)
    for m in modes:
        if m == 0:
            continue
        got, en, ms = run(m)
        d = np.abs(en.view(np.int32).astype(np.int64) - bi)
        moved = int(np.count_nonzero(d))
        ints = {k: int(np.count_nonzero(got[k] != base_got[k])) for k in INTS}
        flts = {k: int(np.count_nonzero(got[k].view(np.int32) != base_got[k].view(np.int32))) for k in FLOATS}
        out["modes"][str(m)] = {
            "env_windows_ms": ms, "speedup_vs_mode0": base_ms / ms if ms else None,
            "energies_moved": moved, "energies_moved_per_song": moved / a.songs,
            "energies_moved_per_window": moved / max(n_windows, 1),
            "max_ulp_moved": int(d.max()) if d.size else 0,
            "int_fields_changed": ints, "float_fields_not_bit_identical": flts,
            "max_abs_attack_diff": float(np.max(np.abs(got["attack"] - base_got["attack"]))),
        }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
