#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_step_fused.py tests/test_gpu_loop_engine.py -x -q -m gpu 2>&1 | tail -5 | tee $O/tests.log
for rep in 1 2; do for f in 0 1; do
  EHM_STEP_FUSED=$f python bench.py --cpu-seconds 0 --no-legs --no-configs > $O/b_${f}_$rep.json 2>$O/err.log
  python - <<P
import json
d=json.loads(open("$O/b_${f}_$rep.json").read().strip().splitlines()[-1])
print("fused=$f:", round(d["value"],1), round(d["ms_per_step"],2), {k:(round(v["ms_per_call"],2)) for k,v in d["breakdown_ms"]["sampling_loop_by_launch_class"].items()}, d["breakdown_ms"].get("encoders_and_projections_once"))
P
done; done
