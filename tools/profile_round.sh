#!/bin/bash
# One round of measurements on the GPU box: bench JSON lines, rocprofv3 kernel stats of the same command, FETCH/WRITE_SIZE of the
# dominant kernel (separate --pmc passes).  Run from the repo root through gpurun; copy what matters from gpurun_out/ to profiles/.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01b; mkdir -p $O
cd $R
timeout 600 python bench.py > $O/bench_ddpm100.json 2> $O/bench_ddpm100.err; tail -1 $O/bench_ddpm100.json | cut -c1-400
timeout 300 python bench.py --workload c2_ddim10 --cpu-seconds 0 > $O/bench_c2.json 2>> $O/bench_ddpm100.err; tail -1 $O/bench_c2.json | cut -c1-200
cd /tmp
MIOPEN_FIND_MODE=FAST timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $O/bench_under_rocprof.json 2> $O/rocprof.err
tail -1 $O/bench_under_rocprof.json | cut -c1-200
ls $O/kt | head; python $R/tools/kstats.py $O/kt | head -25
for c in FETCH_SIZE WRITE_SIZE; do
  MIOPEN_FIND_MODE=FAST timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python $R/tools/bench_hidden.py f16x3 5 > $O/pmc_$c.log 2>&1
done
python $R/tools/pmc_summary.py $O hidden
