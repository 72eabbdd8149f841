cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04m; mkdir -p $O
cd $R
EHM_LOOP_DEBUG=1 timeout 300 python bench.py --workload c2_ddim10 --cpu-seconds 0 --no-legs --steps 2 --warmup 1 > $O/b.json 2> $O/b.err; grep  "^loop:" $O/b.err | cut -c1-250 | head -80
