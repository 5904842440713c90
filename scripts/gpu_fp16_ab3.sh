# fp16 two-plane GEMM: the pipelined in-place DPP exchange (chain) against the scan.  (The 128-row tile
# measured in the same visit -- APS_FP16X2_TM=128, profiles/r02_fp16x2_ab.txt -- is no longer in the source.)
O=gpurun_out/r02_fp16ab3
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "fp16 or chain" > $O/tests.log 2>&1
echo "tests exit $?"; tail -2 $O/tests.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline > $O/joint_$tag.json 2> $O/joint_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/joint_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["frac"], d.get("single_stream_ms_per_step"))
except Exception as e:
    print("$tag failed", e); print(open("$O/joint_$tag.err").read()[-1500:])
PY
}
run scan X=1
run chain APS_GEMM_ROWMAX_CHAIN=1
run scan_again X=1
run chain_again APS_GEMM_ROWMAX_CHAIN=1
(cd /tmp && APS_GEMM_ROWMAX_CHAIN=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_chain1 -o trace -- \
   python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --replicas 1 > $R/$O/bench_under_rocprof_chain1.json 2>&1)
python scripts/trace_by_grid.py $(find $O/prof_chain1 -name "*kernel_trace.csv" | head -1) > $O/by_grid_chain1.txt 2>/dev/null
grep "gemm_fp16x2\|row_exp" $O/by_grid_chain1.txt | head -9 | cut -c1-170
rm -f $O/prof_chain1/*kernel_trace.csv
