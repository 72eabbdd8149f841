cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04n; mkdir -p $O
cd $R
for ib in 2 4; do
echo "ITEM BLOCKS $ib"
EHM_LOOP_ITEM_BLOCKS=$ib EHM_LOOP_DEBUG=1 timeout 300 python tools/loop_try.py 2>&1 | grep "^loop:\|====\|ERR" | cut -c1-160
done
