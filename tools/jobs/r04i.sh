cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O
cd $R
EHM_LOOP_DEBUG=1 timeout 300 python -m pytest "tests/test_gpu_loop_engine.py::test_one_launch_loop_is_bit_equal_to_the_per_step_loop[256-ddim5-True-f16x3-None]" -x -q -s > $O/pytest_engine.txt 2>&1; grep -A16 "^loop:" $O/pytest_engine.txt | cut -c1-400 | head -60; tail -3 $O/pytest_engine.txt
