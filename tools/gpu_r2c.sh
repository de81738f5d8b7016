mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
export T4_TIMING=1
timeout 600 python tools/e2e_time.py 20000 400 1 8 > gpurun_out/r2c_e2e_20k.txt 2>&1; cat gpurun_out/r2c_e2e_20k.txt
timeout 1500 python tools/e2e_time.py 100000 2000 1 8 > gpurun_out/r2c_e2e_100k.txt 2>&1; cat gpurun_out/r2c_e2e_100k.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_stage1_e2e.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r2c_tests.txt
cat gpurun_out/r2c_tests.txt
