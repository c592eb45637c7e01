#!/bin/bash
# every fuzzer of scratch/ once, with the seed given (on the GPU box): scratch/fuzz_all.sh SEED [SCALE]
S=${1:-1}; K=${2:-1}
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/fuzz; mkdir -p $O
rc=0
FUZZ_TAGS=0.5 timeout -k 10 900 python scratch/fuzz_parity.py $((300*K)) $S > $O/all_parity_$S.log 2>&1 || rc=1; tail -1 $O/all_parity_$S.log
timeout -k 10 600 python scratch/fuzz_pnp.py $((4000*K)) $S > $O/all_pnp_$S.log 2>&1 || rc=1; tail -1 $O/all_pnp_$S.log
timeout -k 10 600 python scratch/fuzz_hostpath.py $((150*K)) $S > $O/all_host_$S.log 2>&1 || rc=1; tail -1 $O/all_host_$S.log
timeout -k 10 600 python scratch/fuzz_variants.py $((150*K)) $S > $O/all_var_$S.log 2>&1 || rc=1; tail -1 $O/all_var_$S.log
timeout -k 10 600 python scratch/fuzz_stages.py $((120*K)) $S > $O/all_stg_$S.log 2>&1 || rc=1; tail -1 $O/all_stg_$S.log
timeout -k 10 600 python scratch/fuzz_images.py $((120*K)) $S > $O/all_img_$S.log 2>&1 || rc=1; tail -1 $O/all_img_$S.log
exit $rc
