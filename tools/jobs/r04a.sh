set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O
cd $R
timeout 300 python tools/enc_split.py > $O/enc_split.txt 2>&1; cat $O/enc_split.txt
timeout 300 python bench.py --workload c2_ddim10 --cpu-seconds 0 --no-legs --steps 10 --warmup 3 > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.json | cut -c1-200
cd /tmp
MIOPEN_FIND_MODE=FAST timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --workload c2_ddim10 --steps 5 --warmup 2 --cpu-seconds 0 --no-legs --f16x3-last-steps 10 > $O/c2_under_rocprof.json 2> $O/rocprof.err
python $R/tools/kstats.py $O/kt 60 | tee $O/c2_kstats.txt
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/c2_kernel_stats.csv
cp $(find $O/kt -name "*kernel_trace.csv" | head -1) $O/c2_kernel_trace.csv
rm -rf $O/kt
