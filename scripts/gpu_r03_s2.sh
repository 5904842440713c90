# round 3, session 2: the library without packed-fp32 instructions -- full GPU suite, the disturbance
# again (24 rounds; control: only the STFT built WITH packed ops), STFT at 3 waves / no scratch, a cycle
# trace of the fp16 GEMM's K step, the new bench line
O=gpurun_out/r03_s2
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=$R/aps_amd/csrc
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -25 $O/pytest_gpu.log | cut -c1-250
grep "\[fp16x2\]\|\[joint, batch\|\[config 4" $O/pytest_gpu.log | head -40
for v in nopk pkstft; do
  APS_AMD_LIB=$L/libaps_amd_dist_$v.so APS_GEMM_SPLIT_LAYOUT=1 APS_SPLIT_TM=32 REPLICA_DIFF_ROUNDS=24 timeout 300 python scripts/replica_diff.py 2 4 > $O/diff_$v.log 2>&1
  echo "$v: exit $? reports $(grep -c 'elements differ' $O/diff_$v.log) $(grep 'eager twice\|lstm timeouts' $O/diff_$v.log | tr '\n' ' ')"
done
for spec in "8064 1024 512 ln" "8064 512 512" "8064 2048 512 ln" "31872 2048 512"; do
  APS_AMD_LIB=$L/libaps_amd_trace.so timeout 120 python scripts/gemm_trace.py $spec 2>&1 | tail -24
done > $O/gemm_trace.txt 2>&1
head -30 $O/gemm_trace.txt
fe() { tag=$1; shift; env "$@" timeout 200 python bench.py --workload frontend --no-cpu-baseline > $O/fe_$tag.json 2> $O/fe_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/fe_$tag.json").read().strip().splitlines()[-1])
    sr=d["stage_roofline"]; print("$tag", d["value"], {k:(sr[k]["us_per_launch"], sr[k]["frac"]) for k in ("stft","features","mvdr_weights","beamform")}, sr["all_stages"]["frac"])
except Exception as e:
    print("$tag failed", e); print(open("$O/fe_$tag.err").read()[-1200:])
PY
}
fe default X=1
fe stftlb3 APS_AMD_LIB=$L/libaps_amd_stftlb3.so
fe default_again X=1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/joint_driver_style.json 2> $O/joint_driver_style.err
echo "bench exit $?"; python - <<PY
import json
try:
    d=json.loads(open("$O/joint_driver_style.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["dtype"], d["roofline"]["frac"], d["roofline"]["algorithmic"], d.get("parity"), d.get("replay_checks"), d.get("fp32_path_tiles"))
    b=d.get("baseline_batch"); print("baseline_batch", b and (b["value"], b["ms_per_step"], b["roofline"]["frac"], b["stage_roofline"]["all_stages"]["frac"], b.get("parity")))
    print("cpu", d.get("cpu_baseline"))
except Exception as e:
    print("failed", e); print(open("$O/joint_driver_style.err").read()[-2500:])
PY
