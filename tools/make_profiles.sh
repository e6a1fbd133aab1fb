#!/bin/bash
# Regenerates a round's profile artefacts on the GPU box (run through gpurun from the repo root):
#   kernel_stats_8192songs.csv          rocprofv3 --kernel-trace --stats of the default bench command
#   bench_8192songs_under_rocprof.json  the bench line of that same run (HIP-event kernel times to compare)
#   hbm_traffic.json                    FETCH_SIZE / WRITE_SIZE and SQ counters per kernel at 8192 songs
#                                       (one --pmc pass per counter group, --kernel-trace only)
#   bench_8192songs.json                the bench line of the driver's command (--gpus 1 --steps 20 --warmup 5) outside the
#                                       profiler; *_details.json: the full record bench.py writes beside its compact line
# usage: tools/make_profiles.sh [tag] [songs for the PMC passes]
set -u
TAG=${1:-r06}
PSONGS=${2:-8192}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BDEF="python bench.py --no-cpu-baseline --no-other-configs --no-live-traffic"
BPMC="python bench.py --songs-per-gpu $PSONGS --steps 1 --warmup 0 --no-cpu-baseline --verify 0 --no-mode0-pass --no-other-configs --no-live-traffic"
(cd $ROOT && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BDEF \
   --details-out $OUT/bench_8192songs_under_rocprof_details.json > $OUT/bench_8192songs_under_rocprof.json 2> $OUT/trace.log)
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- $BDEF  (default workload: 8192 songs; MI355X, $TAG)"; cat "$f"; } \
   > $OUT/kernel_stats_8192songs.csv
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
         "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  (cd $ROOT && timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc$i -- $BPMC \
     > $OUT/pmc$i.log 2>&1)
done
python $ROOT/tools/pmc_to_json.py $OUT "$BPMC" $PSONGS > $OUT/hbm_traffic.json
# the driver's own command: the compact line (what BENCH_r*.json parses) and the full record beside it
(cd $ROOT && timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --details-out $OUT/bench_8192songs_details.json \
   > $OUT/bench_8192songs.json 2> $OUT/bench_8192songs.log)
rm -rf $OUT/trace $OUT/pmc[0-9]  # raw traces are large; the summaries above are what gets committed
ls -la $OUT
