cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04p; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_loop_engine.py -x -q > $O/pytest_engine.txt 2>&1; tail -5 $O/pytest_engine.txt
for ib in 2 3 4; do
echo "ITEM BLOCKS $ib"
EHM_LOOP_ITEM_BLOCKS=$ib EHM_LOOP_DEBUG=1 timeout 300 python tools/loop_try.py 2>&1 | grep "^loop:\|====\|ERR" | cut -c1-160
done
