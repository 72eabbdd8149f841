#!/bin/bash
# posefeat_bwd on the exact-f32 MFMA: guidance tests, then C3 A/B against the vector-ALU kernel (EHM_POSEFEAT_VALU=1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_guidance.py tests/test_gpu_configs.py tests/test_gpu_step_fused.py tests/test_gpu_edges.py -x -q -m gpu 2>&1 | tail -5 | tee $O/tests.log
for v in 1 0 1 0; do
  EHM_POSEFEAT_VALU=$v python bench.py --workload c3_guided --cpu-seconds 0 --no-legs --steps 2 > $O/b_c3_$v.json 2>$O/err.log
  python - <<P
import json
d=json.loads(open("$O/b_c3_$v.json").read().strip().splitlines()[-1])
print("c3 posefeat_valu=$v:", round(d["value"],1), round(d["ms_per_step"],2), {k:(round(v["ms_per_call"],2)) for k,v in d["breakdown_ms"]["sampling_loop_by_launch_class"].items() if k.startswith("guid")})
P
done
