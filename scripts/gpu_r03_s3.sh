O=gpurun_out/r03_s3
mkdir -p $O
L=$GRAFT_REPO_ROOT/aps_amd/csrc
for spec in "8064 1024 512 ln" "8064 512 512" "8064 2048 512"; do
  APS_AMD_LIB=$L/libaps_amd_trace.so timeout 120 python scripts/gemm_trace.py $spec 2>&1 | grep -v "per wave\|lgkmcnt(0)  " 
done > $O/gemm_trace.txt 2>&1
cat $O/gemm_trace.txt | cut -c1-250
