# round 3, session 8: the LSTM recurrence on two-plane f16 MFMA against the fp32 build
# (libaps_amd_lstm32.so = the same sources with -DAPS_LSTM_F16=0)
O=gpurun_out/r03_s8
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=$R/aps_amd/csrc
timeout 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_replicas.py tests/test_gpu_dccrn.py tests/test_gpu_joint.py -q -m gpu -k "lstm or dccrn or joint or rnn" > $O/pytest_lstm.log 2>&1
echo "lstm tests exit $?"; tail -15 $O/pytest_lstm.log | cut -c1-220
for n in 32 128; do
  echo "== f16 planes, N=$n"; timeout 120 python scripts/lstm_probe.py $n 249 512 512 2>&1 | grep debug
  echo "== fp32, N=$n"; APS_AMD_LIB=$L/libaps_amd_lstm32.so timeout 120 python scripts/lstm_probe.py $n 249 512 512 2>&1 | grep debug
done
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline > $O/joint_$tag.json 2> $O/joint_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/joint_$tag.json").read().strip().splitlines()[-1])
    b=d.get("baseline_batch",{})
    print("$tag", d["value"], d["ms_per_step"], "single", d.get("single_stream_ms_per_step"), "parity", d["parity"]["enc_out"], "| batch 32:", b.get("value"), b.get("ms_per_step"), b.get("single_stream_ms_per_step"), b.get("stage_us"))
except Exception as e:
    print("$tag failed", e); print(open("$O/joint_$tag.err").read()[-1500:])
PY
}
run f16 X=1
run fp32 APS_AMD_LIB=$L/libaps_amd_lstm32.so
run f16_again X=1
