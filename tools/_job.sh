timeout 1500 python -m pytest tests/test_gpu_guidance.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
bash tools/ab_bench.sh 2 c3_guided base lib_prev.so
