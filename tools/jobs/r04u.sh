cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04u; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv_x2 or resnet or backbone" -s > $O/pytest.txt 2>&1; grep "conv_x2\|passed\|failed\|Error" $O/pytest.txt | tail -14
timeout 300 python tools/enc_layers.py > $O/enc_layers_sk.txt 2>&1; tail -32 $O/enc_layers_sk.txt
EHM_CONV_NO_STREAMK=1 timeout 300 python tools/enc_layers.py > $O/enc_layers_nosk.txt 2>&1; tail -3 $O/enc_layers_nosk.txt
