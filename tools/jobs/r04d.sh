#!/bin/bash
# fused step launch: bit-equality tests, then same-box A/B of the default bench line (EHM_STEP_FUSED=0 = per-step launches)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_step_fused.py -x -q -m gpu 2>&1 | tail -15 | tee $O/tests.log
for rep in 1 2; do
  EHM_STEP_FUSED=0 python bench.py --cpu-seconds 0 --no-legs --no-configs > $O/bench_per_step_$rep.json 2>$O/err.log; tail -1 $O/bench_per_step_$rep.json | cut -c1-160
  EHM_STEP_FUSED=1 python bench.py --cpu-seconds 0 --no-legs --no-configs > $O/bench_fused_$rep.json 2>$O/err.log; tail -1 $O/bench_fused_$rep.json | cut -c1-160
done
python - <<'P'
import json
for f in ("bench_per_step_2","bench_fused_2"):
    d=json.loads(open(f"/root/repo/gpurun_out/r04d/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], {k:(round(v["ms_per_call"],2),v["launches_per_call"]) for k,v in d["breakdown_ms"]["sampling_loop_by_launch_class"].items()})
P
