#!/usr/bin/env python3
"""Folds the per-dispatch counter CSVs of tools/make_profiles.sh into one JSON: HBM bytes per kernel
(FETCH_SIZE / WRITE_SIZE with the gfx950 corrections of MI355X_MICROARCH.md) and the SQ counters.
usage: pmc_to_json.py <dir with pmc*/> "<profiled command>" [songs in the profiled batch]"""
import collections
import csv
import glob
import json
import sys

SONGS = 256
ALGO_BYTES_PER_SONG = 15876000 * 2  # 180 s x 44.1 kHz x 2 ch x s16


def _read(name):
    import os
    try:
        return open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), name)).read().strip()
    except OSError:
        return None


def _fir_mode():
    """the FIR mode the profiled library runs by default (bl_amd_fir_mode), None without the library"""
    import os
    import sys
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bliss_amd
        return int(bliss_amd.load().bl_amd_fir_mode())
    except Exception:
        return None


def main(root, cmd, songs=SONGS):
    global SONGS
    SONGS = int(songs)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in sorted(glob.glob(f"{root}/pmc*/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    algo = SONGS * ALGO_BYTES_PER_SONG
    out = {
        "command": f"rocprofv3 --pmc <group> --kernel-trace --output-format csv -- {cmd}  (one pass per group)",
        "note": "FETCH_SIZE/WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts exactly half of a wide "
                "coalesced streaming read (MI355X_MICROARCH.md, HBM section), so read bytes = 2 * FETCH_SIZE * "
                "1024; WRITE_SIZE taken as is.  One launch per kernel covers all songs.  SQ_* counters: "
                "INSTS/ACTIVE_INST_VALU and WAVE_CYCLES count in units of 4 cycles; VALU busy = "
                "ACTIVE_INST_VALU * 4 / (1024 SIMDs * kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32.",
        "songs": SONGS,
        "git_head": _read(".git_head"),
        "fir_mode": _fir_mode(),
        "algorithmic_bytes_per_launch": algo,
        "kernels": {},
    }
    for k, c in sorted(acc.items()):
        e = {n: v for n, v in sorted(c.items())}
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            rd = 2.0 * c.get("FETCH_SIZE", 0.0) * 1024.0
            wr = c.get("WRITE_SIZE", 0.0) * 1024.0
            e.update(read_bytes_corrected=rd, write_bytes=wr, hbm_bytes=rd + wr,
                     hbm_bytes_per_song=(rd + wr) / SONGS, ratio_to_algorithmic=(rd + wr) / algo)
        if c.get("SQ_BUSY_CYCLES") and c.get("SQ_ACTIVE_INST_VALU"):
            cyc = c["SQ_BUSY_CYCLES"] / 32.0
            e["kernel_cycles"] = cyc
            e["valu_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc)
            if c.get("SQ_LDS_IDX_ACTIVE"):
                e["lds_busy_frac"] = c["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc)
        out["kernels"][k] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:4])
