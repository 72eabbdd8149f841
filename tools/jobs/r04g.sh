#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O
cd $R
for w in c3_guided c2_ddim10; do for f in 0 1 0 1; do
  EHM_STEP_FUSED=$f python bench.py --workload $w --cpu-seconds 0 --no-legs --steps 2 > $O/b_${w}_$f.json 2>$O/err.log
  python - <<P
import json
d=json.loads(open("$O/b_${w}_$f.json").read().strip().splitlines()[-1])
print("$w fused=$f:", round(d["value"],1), round(d["ms_per_step"],2), {k:(round(v["ms_per_call"],2)) for k,v in d["breakdown_ms"]["sampling_loop_by_launch_class"].items()})
P
done; done
python tools/latency_small.py 2>&1 | tail -12
EHM_STEP_FUSED=0 python tools/latency_small.py 2>&1 | tail -12
