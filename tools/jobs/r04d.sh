set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_loop_engine.py -x -q > $O/pytest_engine.txt 2>&1; tail -30 $O/pytest_engine.txt
