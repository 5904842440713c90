# round 3, session 7: everything changed since session 6 (DCCRN train(), pruned kernels, a13 complex
# matmul / inverse, STFT flat-run store, conv2d subsampling on the fp16 kernel, stream pool) under
# the full GPU suite, then the driver-style bench line
O=gpurun_out/r03_s7
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_dccrn_train.py tests/test_gpu_cplx.py -q -m gpu > $O/pytest_new.log 2>&1
echo "new tests exit $?"; tail -40 $O/pytest_new.log | cut -c1-220
timeout 1200 python -m pytest tests -q -m gpu --deselect tests/test_gpu_dccrn_train.py --deselect tests/test_gpu_cplx.py > $O/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -25 $O/pytest_gpu.log | cut -c1-250
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "bench exit $?"
tail -c 3000 $O/bench_driver_style.json
