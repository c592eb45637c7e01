set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r3n; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/fid -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_fid.py 1024 > $O/fid.log 2>&1
