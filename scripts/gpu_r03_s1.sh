# round 3, session 1: GPU tests of the sound fp16 two-plane kernel, build variants of it on the joint
# step, the co-residency disturbance against three STFT builds, one-stream kernel table
O=gpurun_out/r03_s1
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=$R/aps_amd/csrc
timeout 600 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -15 $O/pytest_gpu.log | cut -c1-300
grep "\[fp16x2\]\|\[joint, batch\|\[config 4" $O/pytest_gpu.log | head -40
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline $BENCH_ARGS > $O/joint_$tag.json 2> $O/joint_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/joint_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], "gemm ms", d["roofline"]["kernel_ms_per_step"], "single", d.get("single_stream_ms_per_step"))
except Exception as e:
    print("$tag failed", e); print(open("$O/joint_$tag.err").read()[-1500:])
PY
}
run default X=1
run a1 APS_AMD_LIB=$L/libaps_amd_a1.so
run wjit APS_AMD_LIB=$L/libaps_amd_wjit.so
run wg4 APS_AMD_LIB=$L/libaps_amd_wg4.so
run default_again X=1
BENCH_ARGS="--group 1" run group1 X=1
for v in distA distB distC; do
  APS_AMD_LIB=$L/libaps_amd_$v.so APS_GEMM_SPLIT_LAYOUT=1 APS_SPLIT_TM=32 REPLICA_DIFF_ROUNDS=12 timeout 240 python scripts/replica_diff.py 2 4 > $O/diff_$v.log 2>&1
  echo "$v: exit $? reports $(grep -c 'elements differ' $O/diff_$v.log) $(grep 'eager twice\|lstm timeouts' $O/diff_$v.log | tr '\n' ' ')"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o trace -- \
   python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --replicas 1 > $R/$O/bench_under_rocprof.json 2>&1)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/joint_one_stream_kernel_stats.csv 2>/dev/null; head -14 $f | cut -c1-160
rm -f $O/prof/*/*kernel_trace.csv $O/prof/*kernel_trace.csv
