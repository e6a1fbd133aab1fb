#!/bin/bash
# Kernel durations of the 10 000 x 10 000 bl_distance / bl_cosine_similarity matrices under rocprofv3
# --kernel-trace --stats (HIP events around a 70 us kernel also see the launch gap; the kernel trace does not).
# Runs tools/dist_bench.py (product root variant only) and prints one JSON object.
# usage (through gpurun, from the repo root): tools/dist_prof.sh [reps]
REPS=${1:-50}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/dist_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $ROOT && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -- \
   python tools/dist_bench.py --variants 1 --reps $REPS > $OUT/bench.json 2> $OUT/bench.log)
python - "$OUT" "$REPS" <<'PY'
import csv, glob, json, sys
out, reps = sys.argv[1], int(sys.argv[2])
rows = {}
for f in glob.glob(out + "/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_pairwise" not in k:
            continue
        key = "cosine" if "k_pairwise<true" in k or "ILb1E" in k else "distance"
        rows.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
b = 4 * 10000 * 10000 + 16 * 10000
res = {"tool": "tools/dist_prof.sh: tools/dist_bench.py under rocprofv3 --kernel-trace, N = 10 000, MI355X", "algorithmic_bytes": b}
for k, v in rows.items():
    v = sorted(v)[: max(1, len(v) - 3)]  # drop the three slowest (first launches)
    avg = sum(v) / len(v)
    res[k] = {"launches": len(v), "avg_us": avg, "min_us": v[0], "TBps": b / avg / 1e6, "frac_hbm_peak": b / avg / 1e6 / 8.0}
try:
    res["hip_event_timed"] = json.load(open(out + "/bench.json"))
except Exception as e:
    res["hip_event_timed"] = str(e)
print(json.dumps(res, indent=1))
PY
