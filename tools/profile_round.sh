#!/bin/bash
# One round of measurements on the GPU box: the default bench run (headline + BASELINE configs 2-5 as sub-objects), rocprofv3 kernel stats of the headline
# command, FETCH/WRITE_SIZE of the two hidden-conv chain kernels (separate --pmc passes over tools/bench_hidden.py, relu-like activations).
# Run from the repo root through gpurun:  bash tools/profile_round.sh r03a ; copy what matters from gpurun_out/<tag>/ to profiles/.
TAG=${1:-r03}
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
# stdout of bench.py = the compact line (*_line.json); the full object is bench_detail.json (*_detail.json)
( time timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err ) 2>&1 | grep real; cp bench_detail.json $O/bench_default_detail.json; cat $O/bench_default_line.json
timeout 300 python bench.py --weights insensitive --cpu-seconds 0 --no-legs --no-configs > $O/bench_ddpm100_insensitive_weights_line.json 2>> $O/bench_default.err; cp bench_detail.json $O/bench_ddpm100_insensitive_weights_detail.json; cut -c1-200 $O/bench_ddpm100_insensitive_weights_line.json
cd /tmp
MIOPEN_FIND_MODE=FAST timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-legs --no-configs --f16x3-last-steps $(python -c "import json,sys; print(json.load(open(\"$O/bench_default_detail.json\"))[\"schedule\"][\"f16x3_last_steps\"])") > $O/bench_ddpm100_under_rocprof.json 2> $O/rocprof.err
tail -1 $O/bench_ddpm100_under_rocprof.json | cut -c1-200
python $R/tools/kstats.py $O/kt 40
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/bench_ddpm100_kernel_stats.csv
MIOPEN_FIND_MODE=FAST timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -o kt -- python $R/bench.py --workload c3_guided --steps 2 --warmup 1 --cpu-seconds 0 --no-legs > $O/bench_c3_under_rocprof.json 2>> $O/rocprof.err
cp $(find $O/kt_c3 -name "*kernel_stats.csv" | head -1) $O/bench_c3_guided_kernel_stats.csv
for p in f16 f16x3; do for c in FETCH_SIZE WRITE_SIZE; do
  MIOPEN_FIND_MODE=FAST EHM_STACK=1 EHM_WARMUP=2 timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${p}_$c -o pmc -- python $R/tools/bench_hidden.py $p 5 > $O/pmc_${p}_$c.log 2>&1
done; done
python $R/tools/pmc_traffic.py $O > $O/pmc_traffic.json; cat $O/pmc_traffic.json
rm -rf $O/kt $O/kt_c3
