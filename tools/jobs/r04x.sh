cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04x; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_nonfinite.py -x -q -s > $O/pytest.txt 2>&1; grep -i "stem\|passed\|failed\|Error" $O/pytest.txt | tail -8
timeout 300 python tools/enc_layers.py 2>&1 | grep -i "stem\|total"
EHM_STEM_VALU=1 timeout 300 python tools/enc_layers.py 2>&1 | grep -i "stem\|total"
timeout 300 python tools/enc_split.py 2>&1 | tail -4
timeout 300 python bench.py --workload c2_ddim10 --cpu-seconds 0 --no-legs --steps 10 --warmup 3 2>/dev/null | cut -c1-200
