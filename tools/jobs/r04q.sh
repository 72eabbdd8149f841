cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04q; mkdir -p $O
cd $R
EHM_HIPCC_FLAGS=-DEHM_LOOPSTAT EHM_LIB_PATH=/tmp/libegohmr_stat.so timeout 900 python tools/loop_stats.py ddim20 > $O/loop_stats.json 2> $O/err.txt; cat $O/loop_stats.json; tail -5 $O/err.txt
