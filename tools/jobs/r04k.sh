cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04k; mkdir -p $O
cd $R
EHM_LOOP_DEBUG=1 timeout 300 python -m pytest "tests/test_gpu_loop_engine.py::test_one_launch_loop_without_lbs_every_step_and_unfused_passes" -x -q -s > $O/pytest_engine.txt 2>&1; grep -A16 "^loop:" $O/pytest_engine.txt | cut -c1-300 | head -40; tail -3 $O/pytest_engine.txt
EHM_LOOP_DEBUG=1 timeout 300 python bench.py --workload c2_ddim10 --cpu-seconds 0 --no-legs --steps 2 --warmup 1 > $O/b.json 2> $O/b.err; grep -A14 "^loop:" $O/b.err | cut -c1-250 | head -80
