set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O
cd $R
EHM_HIPCC_FLAGS=-DEHM_LOOPSTAT EHM_LIB_PATH=/tmp/libegohmr_stat.so timeout 900 python tools/loop_stats.py ddim10 > $O/loop_stats_ddim10.json 2> $O/err.txt; cat $O/loop_stats_ddim10.json; tail -5 $O/err.txt
