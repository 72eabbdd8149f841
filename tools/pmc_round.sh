#!/bin/bash
# SQ / GRBM counters of the hidden-conv tile kernel (tools/bench_hidden.py), one rocprofv3 --pmc pass per counter group.
# Run from the repo root through gpurun; the per-kernel averages land in gpurun_out/pmc_round/summary.txt.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_round; mkdir -p $O
for c in "SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT" "SQ_LDS_ACTIVE" "SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS" "TCC_HIT_sum" "TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  MIOPEN_FIND_MODE=FAST timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/$n -o pmc -- python $R/tools/bench_hidden.py f16x3 5 > $O/$n.log 2>&1 || echo "pass $c failed"
done
python $R/tools/pmc_summary.py $O hidden_f16r | tee $O/summary.txt
