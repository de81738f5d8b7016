mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
export T4_TIMING=1
timeout 600 python tools/gpu_phases_add.py $R/trust4_amd/libt4hip_phase.so 3000 > gpurun_out/r2b_phases_add.txt 2>&1; cat gpurun_out/r2b_phases_add.txt
timeout 600 python tools/e2e_time.py 20000 400 1 8 > gpurun_out/r2b_e2e_20k.txt 2>&1; cat gpurun_out/r2b_e2e_20k.txt
timeout 1200 python tools/e2e_time.py 100000 2000 1 8 > gpurun_out/r2b_e2e_100k.txt 2>&1; cat gpurun_out/r2b_e2e_100k.txt
