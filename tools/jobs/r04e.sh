set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O
cd $R
timeout 300 python bench.py --workload c2_ddim10 --cpu-seconds 0 --no-legs --steps 10 --warmup 3 > $O/bench_c2.json 2> $O/bench_c2.err; python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/r04e/bench_c2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['breakdown_ms']))
P
tail -3 $O/bench_c2.err
timeout 600 python bench.py --cpu-seconds 0 --no-legs --steps 5 --warmup 2 > $O/bench_ddpm100.json 2> $O/bench_ddpm100.err; python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/r04e/bench_ddpm100.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['breakdown_ms']))
P
tail -3 $O/bench_ddpm100.err
EHM_NO_ENGINE=1 timeout 600 python bench.py --cpu-seconds 0 --no-legs --steps 5 --warmup 2 > $O/bench_ddpm100_noengine.json 2> $O/bench_ddpm100_ne.err; python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/r04e/bench_ddpm100_noengine.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['breakdown_ms']))
P
