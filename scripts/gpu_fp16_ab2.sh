# fp16 two-plane GEMM, 96-VGPR build: row-maxima chain on / off (joint line + one-stream kernel table),
# against the unconstrained build
O=gpurun_out/r02_fp16ab2
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
run wg5_chain X=1
run wg5_scan APS_GEMM_ROWMAX_CHAIN=0
run wg2_chain APS_AMD_LIB=aps_amd/csrc/libaps_amd_wg2.so
run wg2_scan APS_AMD_LIB=aps_amd/csrc/libaps_amd_wg2.so APS_GEMM_ROWMAX_CHAIN=0
run wg5_chain_again X=1
for c in 1 0; do
(cd /tmp && APS_GEMM_ROWMAX_CHAIN=$c timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_chain$c -o trace -- \
   python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --replicas 1 > $R/$O/bench_under_rocprof_chain$c.json 2>&1)
python scripts/trace_by_grid.py $(find $O/prof_chain$c -name "*kernel_trace.csv" | head -1) > $O/by_grid_chain$c.txt 2>/dev/null
echo "== chain $c"; grep "gemm_fp16x2\|row_exp\|lstm_layer\|attention_small\|conv_split\|conv_mfma" $O/by_grid_chain$c.txt | head -16 | cut -c1-170
rm -f $O/prof_chain$c/*kernel_trace.csv
done
