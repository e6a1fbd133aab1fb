#!/bin/bash
# Collect PMC counters per kernel in separate passes (no trace domains besides --kernel-trace).
# usage: tools/pmc.sh <outdir> "<counters pass 1>" ["<counters pass 2>" ...]
# Each pass: rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python bench.py (256 songs, 1 step)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $OUT
i=0
for C in "$@"; do
  i=$((i+1))
  (cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -- \
     python bench.py --songs-per-gpu 256 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pass$i.log 2>&1)
  f=$(find $OUT/pass$i -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0][:40]
    acc[(k,r['Counter_Name'])]+=float(r['Counter_Value'])
for (k,c),v in sorted(acc.items()):
    if 'env_windows' in k or 'freq_' in k or 'pcm_scan' in k: print(f"{k:42s} {c:28s} {v:.6g}")
PY
done
