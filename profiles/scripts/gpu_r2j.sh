mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
{
echo "== default (waves_per_eu 4 on the two small tiers: 128 VGPRs, 14 spilled)"
T4_LIB=$R/trust4_amd/libt4hip.so python tools/gpu_pass.py 2000000 3
echo "== wpe3 library, default blocks"
T4_LIB=$R/trust4_amd/variants/wpe3/libt4hip.so python tools/gpu_pass.py 2000000 3
echo "== wpe3 library, T0 6 blocks T1 3 blocks"
T4_T0_BLOCKS=6 T4_T1_BLOCKS=3 T4_LIB=$R/trust4_amd/variants/wpe3/libt4hip.so python tools/gpu_pass.py 2000000 3
} > gpurun_out/r2j_ab.txt 2>&1
cat gpurun_out/r2j_ab.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r2j_tests.txt; cat gpurun_out/r2j_tests.txt
