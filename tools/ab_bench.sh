#!/bin/bash
# Same-box A/B of library builds:  bash tools/ab_bench.sh <reps> <workload> <lib A> <lib B> ...   ("base" = the shipped library; others under .ab/)
# alternates the libraries over tools/ab_loop.py (timing only: no calibration, no finite checks - ablation builds may compute garbage).
REPS=$1; WL=$2; shift 2
for rep in $(seq $REPS); do for lib in "$@"; do
  if [ "$lib" = base ]; then unset EHM_LIB_PATH; else export EHM_LIB_PATH=$GRAFT_REPO_ROOT/.ab/$lib; fi
  timeout 600 python tools/ab_loop.py $WL 5 2>/dev/null | tail -1
done; done
unset EHM_LIB_PATH
