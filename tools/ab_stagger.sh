#!/bin/bash
# A/B of the chain kernel's block phase offset (EgoHMR.chain_stagger / ehm_gcn_set_chain_stagger) on the production launch: the headline bench command,
# alternated over the settings, same box.  bash tools/ab_stagger.sh <tag> "0 2 4 6 0 4"
TAG=${1:-stagger}; SET=${2:-"0 2 4 6 8 0 4"}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
for s in $SET; do
  timeout 300 python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-legs --no-configs --f16x3-last-steps 100 --chain-stagger $s > $O/line_$s.json 2>> $O/err.txt
  python - "$O/line_$s.json" $s <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"stagger {sys.argv[2]:>2}: {d['value']:8.1f} bodies/s  {d['ms_per_step']:7.2f} ms  chain launch {d['roofline']['avg_launch_ms']*1e3:7.1f} us  frac {d['roofline']['frac']}")
PY
done | tee $O/summary.txt
