cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout 900 python -m pytest tests/test_gpu_schedule.py -x -q -s -m gpu 2>&1 | tail -40 > gpurun_out/r03a/sched_tests.log
for g in 0 0.3 0.8 1.0; do
  timeout 600 python tools/precision_schedule.py --batch 256 --T 100 --ks 0,4,8,16,24,32,48,64,80 --seeds 0 --gain $g > gpurun_out/r03a/sched_ddpm100_gain$g.jsonl 2>gpurun_out/r03a/sched_err_$g.log
done
timeout 600 python tools/precision_schedule.py --batch 256 --T 100 --respacing ddim10 --ks 0,2,3,4,5,6,8 --seeds 0 --gain 1.0 > gpurun_out/r03a/sched_ddim10_gain1.jsonl 2>>gpurun_out/r03a/sched_err_1.0.log
timeout 600 python bench.py --cpu-seconds 0 > gpurun_out/r03a/bench_sens.json 2> gpurun_out/r03a/bench_sens.err
tail -c 3000 gpurun_out/r03a/sched_tests.log
