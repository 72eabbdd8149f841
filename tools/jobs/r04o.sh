cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04o; mkdir -p $O
cd $R
EHM_LOOP_DEBUG=1 timeout 300 python tools/loop_try.py 2>&1 | grep "^loop:\|====\|ERR\|first timeout\|tickets\|^  in \|^  out\|^  body" | cut -c1-260
