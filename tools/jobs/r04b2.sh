cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b2; mkdir -p $O
MIOPEN_FIND_MODE=FAST timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-legs --no-configs --f16x3-last-steps 100 > $O/bench_ddpm100_under_rocprof.json 2> $O/rocprof.err
python $R/tools/kstats.py $O/kt 24
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/bench_ddpm100_kernel_stats.csv
rm -rf $O/kt
cd $R
for e in 0 1; do EHM_LOOP_ENGINE=$e timeout 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-legs --no-configs > $O/bench_engine_$e.json 2>/dev/null; python - <<P
import json
d=json.loads(open('/root/repo/gpurun_out/r04b2/bench_engine_$e.json').read().strip().splitlines()[-1])
print('engine', $e, d['value'], d['ms_per_step'], {k:round(v['ms_per_call'],2) for k,v in d['breakdown_ms']['sampling_loop_by_launch_class'].items()})
P
done
