cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04t; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_schedule.py tests/test_gpu_loop_engine.py -x -q -s > $O/pytest.txt 2>&1; grep "^\[g1\|passed\|failed\|Error" $O/pytest.txt | tail -12
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; tail -3 $O/bench_default.err; python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/r04t/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['dtype'][:80])
for k in ('all_steps_f16x3','f32_mfma_path','f16_denoiser_path','schedule_at_contract_tol','default_path_on_insensitive_weights'):
    v=d.get(k); print(k, v and {kk:v[kk] for kk in v if kk in ('value','ms_per_step','vs_default_path','calibrated_f16x3_last_steps')})
for k,v in d['configs'].items(): print(k, v['value'], v['ms_per_step'], list(v.keys()))
print(json.dumps(d['configs']['c3_guided'].get('roofline_guidance'))[:900])
print(json.dumps(d['roofline_hbm'])[:1200])
P
