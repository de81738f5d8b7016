mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > gpurun_out/r02_smoke.txt; cat gpurun_out/r02_smoke.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r02_gpu_tests.txt; cat gpurun_out/r02_gpu_tests.txt
timeout 1500 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; head -c 900 gpurun_out/r02_bench.json; echo
