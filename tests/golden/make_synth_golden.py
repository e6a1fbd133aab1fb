#!/usr/bin/env python3
"""Generates tests/golden/synth_golden.json: expected outputs of the CPU oracle on a few
seeded synthetic songs (the integer generator of oracle/orc_synth.c makes the PCM
byte-identical everywhere; its MD5 is part of the fixture).  The oracle itself is pinned on
the reference's own goldens (tests/test_oracle_golden.py); this fixture freezes its outputs
so that a drift of the oracle or of the platform's libm shows up as a diff, and gives the
GPU tests committed expectations.  Run from the repo root: python tests/golden/make_synth_golden.py"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.oracle_py import Oracle  # noqa: E402

CASES = [  # seed, rate, channels, seconds, extra samples
    (101, 22050, 2, 11, 0), (102, 44100, 2, 30, 0), (103, 44100, 1, 25, 123),
    (104, 22050, 1, 8, 511), (105, 48000, 2, 17, 2), (106, 8000, 1, 2, 0),
    (107, 44100, 2, 180, 0),      # S180: the shape BASELINE.json's metric is quoted on
    (108, 22050, 2, 600, 0),      # a 10-minute song (longest length of BASELINE configs[4])
]


def main():
    orc = Oracle()
    out = {"generator": "oracle/orc_synth.c orc_synth_fill(seed, rate, channels)", "cases": []}
    for seed, rate, ch, secs, extra in CASES:
        n = rate * ch * secs + extra
        pcm = orc.synth(seed, rate, ch, n)
        r = orc.analyze(pcm, ch, secs)
        _, en = orc.envelope(pcm, secs)
        nw = r["n_windows"]
        out["cases"].append({
            "seed": seed, "rate": rate, "channels": ch, "duration": secs, "n_samples": n,
            "pcm_md5": hashlib.md5(pcm.tobytes()).hexdigest(),
            "expect": {k: (float(v) if isinstance(v, float) else int(v)) for k, v in r.items()},
            "energies_first8": [float(x) for x in en[:8]],
            "energies_last8": [float(x) for x in en[max(nw - 8, 0):nw]],
            "energies_md5": hashlib.md5(en[:nw].tobytes()).hexdigest(),
        })
    with open(os.path.join(ROOT, "tests", "golden", "synth_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
