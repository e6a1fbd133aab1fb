#!/bin/bash
# One GPU-box session of a round: GPU tests, the default bench line, the host-path and mixed-length runs.
# usage (through gpurun, from the repo root): tools/round_run.sh <tag> [steps...]
#   steps: tests bench host mixed resample pipeline
set -u
TAG=${1:-r02}; shift || true
STEPS=${*:-tests bench host mixed}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for S in $STEPS; do
  case $S in
    tests) timeout 2400 python -m pytest tests -m gpu -x -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log;;
    bench) timeout 900 python bench.py > $OUT/bench_8192songs.json 2> $OUT/bench_8192songs.log; tail -c 600 $OUT/bench_8192songs.json;;
    host)  timeout 600 python tools/host_path_bench.py --songs 2048 --latency > $OUT/host_path_staged.json 2> $OUT/host_path_staged.log; cat $OUT/host_path_staged.json
           timeout 600 python tools/host_path_bench.py --songs 1024 --mode registered > $OUT/host_path_registered.json 2> $OUT/host_path_registered.log; cat $OUT/host_path_registered.json;;
    mixed) timeout 600 python tools/mixed_bench.py > $OUT/mixed_8192songs.json 2> $OUT/mixed.log; cat $OUT/mixed_8192songs.json;;
    resample) timeout 900 python tools/resample_bench.py --songs 1024 > $OUT/resample.json 2> $OUT/resample.log; cat $OUT/resample.json
              timeout 900 python tools/resample_soak.py --calls 240 > $OUT/resample_soak.json 2>> $OUT/resample.log; cat $OUT/resample_soak.json;;
    pipeline) for R in 44100 48000; do timeout 900 python tools/pipeline_bench.py --songs 4096 --rate $R 2>> $OUT/pipeline.log | tail -1; done > $OUT/pipeline.json; cat $OUT/pipeline.json;;
  esac
done
