#!/bin/bash
# new tests of this commit, then the slab-order A/B of the chain kernels (EHM_CHAIN_SLAB_GROUPS) with the L2-side traffic of each order
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py -x -q -m gpu -k "timestep_vectors or module_calls or resnet50 or hidden_stack" 2>&1 | tail -5 | tee $O/tests.log
for rep in 1 2 3; do for g in 1 2 4 8; do
  echo "f16x3 groups=$g rep=$rep: $(EHM_CHAIN_SLAB_GROUPS=$g EHM_STACK=1 python tools/bench_hidden.py f16x3 300 | tail -1)" | tee -a $O/slab_order_ab.txt
done; done
for g in 1 2 4; do
  echo "f16 groups=$g: $(EHM_CHAIN_SLAB_GROUPS=$g EHM_STACK=1 python tools/bench_hidden.py f16 300 | tail -1)" | tee -a $O/slab_order_ab.txt
done
cd /tmp
for g in 2 4; do mkdir -p $O/g$g; for c in FETCH_SIZE WRITE_SIZE; do
  EHM_CHAIN_SLAB_GROUPS=$g EHM_STACK=1 EHM_WARMUP=2 timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/g$g/pmc_f16x3_$c -o pmc -- python $R/tools/bench_hidden.py f16x3 5 > $O/g$g/pmc_f16x3_$c.log 2>&1
done; python $R/tools/pmc_traffic.py $O/g$g > $O/pmc_traffic_groups$g.json; cat $O/pmc_traffic_groups$g.json; rm -rf $O/g$g; done
