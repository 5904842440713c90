# round 3, session 13: trainable mel filters, the power spectrum inside differentiable feature chains
O=gpurun_out/r03_s13
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py tests/test_gpu_joint.py -q -m gpu -s -k "mel or pow or joint or backward" > $O/pytest_new.log 2>&1
echo "tests exit $?"; grep -E "^\[grad\].*(fbank|filters|abs-pow)|passed|failed|Error|error|^E " $O/pytest_new.log | cut -c1-200 | tail -40
