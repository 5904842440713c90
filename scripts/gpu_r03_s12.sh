# round 3, session 12: backward of XL attention, context windows, the causal conformer convolution and
# the linear / conv1d projections
O=gpurun_out/r03_s12
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -s -k "xl or causal or convolution_module or attention or projection" > $O/pytest_new.log 2>&1
echo "new tests exit $?"; grep -E "^\[grad\]|passed|failed|Error|error|^E " $O/pytest_new.log | cut -c1-200 | grep -v "^\[grad\] \(conv1d\|linear\|causal conv\|conv module\)" | tail -60
