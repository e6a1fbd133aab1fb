#!/bin/bash
# One GPU-box session at the end of a round: the GPU suite, the profile artefacts (tools/make_profiles.sh), the soaks
# against the CPU oracle, the clock probe and the mixed-length run.  Everything lands under gpurun_out/<tag>_final/.
# usage (through gpurun, from the repo root): tools/round_run.sh <tag> [steps...]
#   steps: tests profiles soaks clock mixed
set -u
TAG=${1:-r06}; shift || true
STEPS=${*:-tests profiles soaks clock mixed}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_final
mkdir -p $OUT
cd $ROOT
for S in $STEPS; do
  case $S in
    tests) timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log;;
    profiles) bash tools/make_profiles.sh $TAG 8192 > $OUT/make_profiles.log 2>&1; cp gpurun_out/prof_$TAG/* $OUT/ 2>/dev/null; tail -c 400 $OUT/bench_8192songs.json; echo;;
    soaks) timeout 900 python tools/soak.py --seconds 180 --rate 44100 --stereo --songs 1024 --seed 71 > $OUT/soak_s180_seed71.json 2> $OUT/soak1.log
           timeout 900 python tools/soak.py --songs 8192 --seed 72 --max-seconds 60 > $OUT/soak_8192songs_seed72.json 2> $OUT/soak2.log
           timeout 1200 python tools/soak.py --songs 2048 --seed 73 --max-seconds 600 > $OUT/soak_2048songs_upto600s_seed73.json 2> $OUT/soak3.log
           timeout 900 python tools/soak.py --seconds 180 --rate 44100 --stereo --songs 1024 --seed 74 --fir-mode 0 > $OUT/soak_s180_seed74_fir_mode0.json 2> $OUT/soak4.log
           for f in $OUT/soak_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], d['songs'], 'int mismatches', d['n_int_mismatches'], 'floats out', d['n_float_out_of_tolerance'], 'strict', d.get('n_songs_failing_strict_1e-4_rel'))"; done;;
    clock) timeout 300 python tools/clock_probe.py > $OUT/clock_probe.json 2> $OUT/clock.log; python -c "import json; d=json.load(open('$OUT/clock_probe.json')); print({k:(v.get('sclk_mhz',{}).get('mean'), v.get('power_w',{}).get('mean'), v.get('kernel_ms')) for k,v in d.items() if isinstance(v,dict)})";;
    mixed) timeout 600 python tools/mixed_bench.py > $OUT/mixed_8192songs.json 2> $OUT/mixed.log; cat $OUT/mixed_8192songs.json;;
  esac
done
