set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_loop_engine.py -x -q > $O/pytest_engine.txt 2>&1; tail -5 $O/pytest_engine.txt
for w in c2_ddim10 ddpm100; do
timeout 600 python bench.py --workload $w --cpu-seconds 0 --no-legs --steps 5 --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err; python - <<P
import json
d=json.loads(open('/root/repo/gpurun_out/r04f/bench_$w.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['breakdown_ms']))
P
done
