cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04v; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv_x2 or resnet or backbone" -s > $O/pytest.txt 2>&1; grep "18, 56\|passed\|failed\|Error" $O/pytest.txt | tail -5
timeout 300 python tools/enc_layers.py > $O/enc_layers_sk.txt 2>&1; tail -1 $O/enc_layers_sk.txt
EHM_CONV_NO_STREAMK=1 timeout 300 python tools/enc_layers.py > $O/enc_layers_nosk.txt 2>&1; tail -1 $O/enc_layers_nosk.txt
timeout 300 python tools/enc_split.py 2>&1 | tail -5
