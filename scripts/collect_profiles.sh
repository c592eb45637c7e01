#!/bin/bash
# Runs on the GPU box (gpurun): kernel stats of bench.py + the PMC passes; everything lands under gpurun_out/prof_$TAG.
# usage: scripts/collect_profiles.sh TAG
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# the profiled run skips the host-input / batch-1 / CPU legs: they launch the same kernels on 1 .. 32 frames and would
# drag the per-kernel averages of the summary away from the 1024-frame launches the roofline figures are about
rocprofv3 --kernel-trace --stats -d $O/bench -o bench --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-extra-legs --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cd $R && python3 bench.py > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/dense_$c -o p --output-format csv -- python3 $R/scripts/prof_dense.py 512 > $O/dense_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c -d $O/ingest_$c -o p --output-format csv -- python3 $R/scripts/prof_ingest.py 256 > $O/ingest_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/dense_sq1 -o p --output-format csv -- python3 $R/scripts/prof_dense.py 512 > $O/dense_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY -d $O/dense_sq2 -o p --output-format csv -- python3 $R/scripts/prof_dense.py 512 > $O/dense_sq2.log 2>&1
# the ingest pass's instruction mix (round 4: it runs with its vector pipes ~70 % busy AND at ~87 % of the achievable HBM rate)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/ingest_sq1 -o p --output-format csv -- python3 $R/scripts/prof_ingest.py 256 > $O/ingest_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY -d $O/ingest_sq2 -o p --output-format csv -- python3 $R/scripts/prof_ingest.py 256 > $O/ingest_sq2.log 2>&1
cd $R
python3 scripts/pmc_table.py $O/ingest_sq1 k_ingest > $O/ingest_sq.txt
python3 scripts/pmc_table.py $O/ingest_sq2 k_ingest >> $O/ingest_sq.txt
python3 scripts/summarize_pmc.py $O/dense_FETCH_SIZE/p_counter_collection.csv $O/dense_WRITE_SIZE/p_counter_collection.csv 512 $TAG "k_dense_band<0" 4147200 k_calib_copy_x4 dense > $O/dense_traffic.txt
python3 scripts/summarize_pmc.py $O/dense_FETCH_SIZE/p_counter_collection.csv $O/dense_WRITE_SIZE/p_counter_collection.csv 512 $TAG k_dense_wave 2203200 k_calib_copy_x4 dense_step > $O/dense_step_traffic.txt
python3 scripts/summarize_pmc.py $O/ingest_FETCH_SIZE/p_counter_collection.csv $O/ingest_WRITE_SIZE/p_counter_collection.csv 256 $TAG k_ingest_staged 8294400 > $O/ingest_traffic.txt
python3 scripts/summarize_sq.py $O/dense_sq1 512 $TAG "k_dense_band<0" dense > $O/issue_dense.txt
python3 scripts/summarize_sq.py $O/dense_sq1 512 $TAG k_dense_wave dense_step > $O/issue_dense_step.txt
python3 scripts/pmc_table.py $O/dense_sq1 k_dense > $O/dense_sq.txt
python3 scripts/pmc_table.py $O/dense_sq2 k_dense >> $O/dense_sq.txt
cp profiles/${TAG}_pmc_dense.json profiles/${TAG}_pmc_dense_step.json profiles/${TAG}_pmc_ingest.json profiles/traffic_dense.json profiles/traffic_dense_step.json profiles/traffic_ingest.json profiles/issue_dense.json profiles/issue_dense_step.json $O/ 2>/dev/null || true
tail -1 $O/bench.json
