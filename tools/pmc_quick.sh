#!/bin/bash
# Per-kernel SQ counters of one bench step (separate --pmc passes, --kernel-trace only), printed as a table.
# usage (through gpurun): tools/pmc_quick.sh <songs> [kernel-name-substring ...]
SONGS=${1:-2048}; shift
PAT=${*:-freq_scan freq_frames env_windows pcm_scan}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcq
rm -rf $OUT; mkdir -p $OUT
# BL_AMD_NO_SIDE (the envelope tail serialised behind the other kernels) exists in the measurement build only
make -s -C $ROOT/bliss_amd/csrc measure
export BLISS_AMD_LIB=$ROOT/bliss_amd/libbliss_amd_measure.so
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
         "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
         "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
         "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" \
         "SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_FLAT"; do
  i=$((i+1))
  (cd $ROOT && BL_AMD_NO_SIDE=1 timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- \
     python bench.py --songs-per-gpu $SONGS --steps 1 --warmup 0 --no-cpu-baseline --verify 0 --no-mode0-pass --no-other-configs --no-live-traffic > $OUT/p$i.log 2>&1)
done
python - $OUT $PAT <<'PY'
import csv,sys,glob,collections
root=sys.argv[1]; pats=sys.argv[2:]
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(root+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,c in sorted(acc.items()):
    if not any(p in k for p in pats): continue
    print(k)
    for n,v in sorted(c.items()): print(f"   {n:28s}{v:.5g}")
    if c.get('SQ_BUSY_CYCLES') and c.get('SQ_ACTIVE_INST_VALU'):
        cyc=c['SQ_BUSY_CYCLES']/32.0
        print(f"   kernel_cycles {cyc:.4g}  valu_busy {c['SQ_ACTIVE_INST_VALU']*4/(1024*cyc):.3f}"
              + (f"  lds_busy {c['SQ_LDS_IDX_ACTIVE']/(256*cyc):.3f}" if c.get('SQ_LDS_IDX_ACTIVE') else "")
              + (f"  wait_any/wave_cycles {c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.3f}" if c.get('SQ_WAIT_INST_ANY') and c.get('SQ_WAVE_CYCLES') else ""))
PY
