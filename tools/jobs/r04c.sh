set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O
cd $R
timeout 600 python tools/exp_encoder_precision.py ddim10 > $O/enc_precision_ddim10.jsonl 2> $O/enc_precision.err; cat $O/enc_precision_ddim10.jsonl; tail -3 $O/enc_precision.err
timeout 900 python tools/exp_encoder_precision.py ddpm100 > $O/enc_precision_ddpm100.jsonl 2>> $O/enc_precision.err; cat $O/enc_precision_ddpm100.jsonl
cd /tmp
MIOPEN_FIND_MODE=FAST timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --workload c2_ddim10 --steps 5 --warmup 2 --cpu-seconds 0 --no-legs --f16x3-last-steps 10 > $O/c2_under_rocprof.json 2> $O/rocprof.err
python $R/tools/kstats.py $O/kt 30 | tee $O/c2_kstats.txt
cp $(find $O/kt -name "*kernel_trace.csv" | head -1) $O/c2_kernel_trace.csv
rm -rf $O/kt
