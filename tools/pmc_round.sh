#!/bin/bash
# SQ / GRBM / TCC counters of the two chained hidden-conv kernels (tools/bench_hidden.py with EHM_STACK=1), one rocprofv3 --pmc pass per
# counter.  Run from the repo root through gpurun:  bash tools/pmc_round.sh r02_pmc ; the per-kernel averages land in
# gpurun_out/<tag>/summary.txt (copy to profiles/).
TAG=${1:-pmc_round}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
for p in f16 f16x3; do
for c in "SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS" "TCC_HIT_sum" "TCC_MISS_sum"; do
  MIOPEN_FIND_MODE=FAST EHM_STACK=1 EHM_WARMUP=2 timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${p}_$c -o pmc -- python $R/tools/bench_hidden.py $p 5 > $O/${p}_$c.log 2>&1 || echo "pass $p $c failed"
done
done
python $R/tools/pmc_summary.py $O gcn_hidden_chain | tee $O/summary.txt
for p in f16 f16x3; do python $R/tools/pmc_clock.py $O/${p}_GRBM_GUI_ACTIVE gcn_hidden_chain | sed "s/^/$p: /" | tee -a $O/summary.txt; done
