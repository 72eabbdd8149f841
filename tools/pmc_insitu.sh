#!/bin/bash
# SQ / GRBM counters of the chain kernel ON THE PRODUCTION LAUNCH: one rocprofv3 --pmc pass per counter over the headline bench command
# (every step split-f16, 100-step DDPM at B = 256, the sampler's own launches - not tools/bench_hidden.py), plus a --kernel-trace pass for the
# durations of the same launches and tools/power_probe_bench.py sampling socket power / shader clock while that loop runs.
# Run from the repo root through gpurun:  bash tools/pmc_insitu.sh r06a_insitu ; the summary lands in gpurun_out/<tag>/summary.txt.
TAG=${1:-pmc_insitu}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
CMD="python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-legs --no-configs --f16x3-last-steps 100"
for c in "SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY" "SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  MIOPEN_FIND_MODE=FAST timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o pmc -- $CMD > $O/$c.log 2>&1 || echo "pass $c failed"
  tail -1 $O/$c.log | cut -c1-160
done
python $R/tools/pmc_summary.py $O gcn_hidden_chain | tee $O/summary.txt
python $R/tools/pmc_clock.py $O/GRBM_GUI_ACTIVE gcn_hidden_chain | tee -a $O/summary.txt
python $R/tools/pmc_insitu_derive.py $O | tee -a $O/summary.txt
# socket power / clock while the production loop runs (no profiler attached)
timeout 300 python $R/tools/power_probe_bench.py 12 > $O/power_probe_bench.jsonl 2> $O/power_probe_bench.err; cat $O/power_probe_bench.jsonl | tee -a $O/summary.txt
# keep the merged output small
find $O -name "*.csv" -size +2M -delete
