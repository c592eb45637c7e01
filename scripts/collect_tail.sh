#!/bin/bash
# Runs on the GPU box (gpurun): SQ counters + kernel stats of the stages AFTER the threshold+corner pass (list, sub-pixel,
# lattice + pose; fiducial quads + tag poses) -- the "tail" of the step.  Lands under gpurun_out/tail_$TAG.
# usage: scripts/collect_tail.sh TAG [NFRAMES]
set -e
TAG=${1:-r03}
N=${2:-512}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/tail_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
SQ2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY"
SQ3="SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_F64 GRBM_GUI_ACTIVE GRBM_COUNT"
for w in board:prof_pnp.py fid:prof_fid.py; do
  name=${w%%:*}; script=${w##*:}
  rocprofv3 --kernel-trace --stats -d $O/${name}_stats -o s --output-format csv -- python3 $R/scripts/$script $N > $O/${name}_stats.log 2>&1
  i=1
  for set in "$SQ1" "$SQ2" "$SQ3"; do
    rocprofv3 --kernel-trace --pmc $set -d $O/${name}_sq$i -o p --output-format csv -- python3 $R/scripts/$script $N > $O/${name}_sq$i.log 2>&1 || echo "counter set $i failed for $name (see log)"
    i=$((i+1))
  done
  : > $O/${name}_sq.txt
  for i in 1 2 3; do python3 $R/scripts/pmc_table.py $O/${name}_sq$i k_ >> $O/${name}_sq.txt 2>/dev/null || true; done
  f=$(find $O/${name}_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${name}_kernel_stats.csv
done
echo "frames per launch: $N" > $O/README.txt
ls $O
