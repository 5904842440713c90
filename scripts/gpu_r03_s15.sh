# round 3, session 15: the transformer decoder under autograd / in train() mode
O=gpurun_out/r03_s15
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_decoder.py -q -m gpu -s -k "decoder or cross_attention or xfmr_asr" > $O/pytest_new.log 2>&1
echo "tests exit $?"; grep -E "^\[grad\].*(asr@|decoder)|passed|failed|Error|error|^E " $O/pytest_new.log | cut -c1-220 | tail -40
