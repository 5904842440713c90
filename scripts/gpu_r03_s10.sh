# round 3, session 10: team form with the one-round-trip slow path and the late request
O=gpurun_out/r03_s10
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_replicas.py tests/test_gpu_dccrn.py tests/test_gpu_joint.py tests/test_variant_rnn.py tests/test_concat_encoder.py -q -m gpu -k "lstm or dccrn or joint or rnn" > $O/pytest_lstm.log 2>&1
echo "lstm tests exit $?"; tail -5 $O/pytest_lstm.log | cut -c1-220
APS_AMD_LIB=$GRAFT_REPO_ROOT/aps_amd/csrc/libaps_amd_lstmtrace.so timeout 200 python scripts/lstm_trace.py 128 249 512 512 2>&1 | grep -A10 "DEBUG=0" | head -12
for n in 16 32 128; do
  echo "== team, N=$n"; timeout 120 python scripts/lstm_probe.py $n 249 512 512 2>&1 | grep "debug=0"
  echo "== spread, N=$n"; APS_LSTM_TEAM=0 timeout 120 python scripts/lstm_probe.py $n 249 512 512 2>&1 | grep "debug=0"
done
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline $EXTRA > $O/joint_$tag.json 2> $O/joint_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/joint_$tag.json").read().strip().splitlines()[-1])
    b=d.get("baseline_batch",{})
    print("$tag", d["value"], d["ms_per_step"], "single", d.get("single_stream_ms_per_step"), "| batch 32:", b.get("value"), b.get("ms_per_step"), b.get("single_stream_ms_per_step"), (b.get("stage_us") or {}).get("mask_net"))
except Exception as e:
    print("$tag failed", e); print(open("$O/joint_$tag.err").read()[-1500:])
PY
}
run team X=1
run spread APS_LSTM_TEAM=0
EXTRA="--replicas 1 --no-baseline-batch"
run team_one_stream X=1
run spread_one_stream APS_LSTM_TEAM=0
