cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04r; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_loop_engine.py -x -q > $O/pytest_engine.txt 2>&1; tail -5 $O/pytest_engine.txt
for ib in 1 2; do
echo "ITEM BLOCKS $ib"
EHM_LOOP_ITEM_BLOCKS=$ib EHM_LOOP_DEBUG=1 timeout 300 python tools/loop_try.py 2>&1 | grep "^loop:\|====\|ERR" | cut -c1-160
done
EHM_HIPCC_FLAGS=-DEHM_LOOPSTAT EHM_LIB_PATH=/tmp/libegohmr_stat.so timeout 900 python tools/loop_stats.py ddim20 > $O/loop_stats.json 2> $O/err.txt; cat $O/loop_stats.json; tail -5 $O/err.txt
