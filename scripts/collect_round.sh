#!/bin/bash
# Runs on the GPU box (gpurun): the whole measurement set of a round -- scripts/collect_profiles.sh (headline bench under
# rocprofv3 --kernel-trace --stats, unprofiled bench, PMC traffic + SQ counters of the dense and ingest kernels),
# scripts/collect_tail.sh (SQ counters + kernel stats of the tail kernels), the configs[3] / configs[4]-style bench lines WITH
# their cpu_baseline and accuracy legs, and the RCCL path with a world of one.  Everything lands under gpurun_out/.
# usage: scripts/collect_round.sh TAG
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round_$TAG
mkdir -p $O
cd $R
bash scripts/collect_profiles.sh $TAG > $O/collect_profiles.log 2>&1 || echo "collect_profiles failed (see log)"
echo "profiles done"
bash scripts/collect_tail.sh $TAG 1024 > $O/collect_tail.log 2>&1 || echo "collect_tail failed (see log)"
echo "tail done"
python3 bench.py --width 3840 --height 2160 --batch 256 --fisheye > $O/config4_fisheye4k_bench.json 2> $O/config4.err || echo "config4 bench failed"
echo "config4 done"
python3 bench.py --fiducials 6x4 > $O/config5_fiducials_bench.json 2> $O/config5.err || echo "config5 bench failed"
python3 bench.py --fiducials 6x4 --tag-refine subpix --no-extra-legs > $O/config5_fiducials_subpix_bench.json 2> $O/config5s.err || echo "config5 (subpix) bench failed"
echo "config5 done"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline --no-extra-legs > $O/world1_rccl_bench.json 2> $O/world1.err || echo "world-1 RCCL bench failed"
echo "world1 done"
ls $O
