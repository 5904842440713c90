O=gpurun_out/r03_s6
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=$R/aps_amd/csrc
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -8 $O/pytest_gpu.log | cut -c1-250
grep "\[fp16x2\]\|\[joint, batch\|\[config 4" $O/pytest_gpu.log | head -40
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-baseline-batch $BENCH_ARGS > $O/joint_$tag.json 2> $O/joint_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/joint_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], "gemm ms", d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"], "single", d.get("single_stream_ms_per_step"), "checks", d.get("replay_checks"))
except Exception as e:
    print("$tag failed", e); print(open("$O/joint_$tag.err").read()[-1500:])
PY
}
run default X=1
run persist APS_AMD_LIB=$L/libaps_amd_persist.so
run wjit0 APS_AMD_LIB=$L/libaps_amd_wjit0.so
run default_again X=1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o trace -- \
   python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-baseline-batch --replicas 1 > $R/$O/bench_under_rocprof.json 2>&1)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/joint_one_stream_kernel_stats.csv 2>/dev/null; head -9 $f | cut -c1-180
rm -f $O/prof/*/*kernel_trace.csv $O/prof/*kernel_trace.csv
