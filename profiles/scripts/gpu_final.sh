# One short GPU session, most important artefacts first (the call may be cut by the budget): smoke, rocprofv3 kernel stats of
# the bench command, the full bench line, the -m gpu tests, then the PMC passes. Outputs under gpurun_out/.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
T0=$(date +%s)
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
echo "smoke done $(( $(date +%s) - T0 )) s"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r01 -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e-pairs 0 > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err)
find $R/gpurun_out/prof -name "*kernel_stats*" | head -1 | xargs -I{} cp {} $R/gpurun_out/kernel_stats.csv
head -9 gpurun_out/kernel_stats.csv; tail -1 gpurun_out/prof_bench.json | cut -c1-400
echo "stats done $(( $(date +%s) - T0 )) s"
timeout 400 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.json
echo "bench done $(( $(date +%s) - T0 )) s"
timeout 600 python -m pytest tests -m gpu -q -n 4 --durations=15 > gpurun_out/pytest_gpu.log 2>&1; tail -25 gpurun_out/pytest_gpu.log
echo "pytest done $(( $(date +%s) - T0 )) s"
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -o pmc -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --e2e-pairs 0 > /dev/null 2>&1
  F=$(find $R/gpurun_out/pmc_$C -name "*counter_collection.csv" | head -1); [ -n "$F" ] && cp $F $R/gpurun_out/pmc_$C.csv
done
[ -f $R/gpurun_out/pmc_FETCH_SIZE.csv ] && [ -f $R/gpurun_out/pmc_WRITE_SIZE.csv ] && python $R/tools/pmc_summary.py $R/gpurun_out/pmc_FETCH_SIZE.csv $R/gpurun_out/pmc_WRITE_SIZE.csv $R/gpurun_out/pmc_summary.json
echo "pmc done $(( $(date +%s) - T0 )) s"
# host phases on the device (opt-in switches) and the k-mer kernels: whole stage 1 with and without, the counting benchmark
cd $R
python tools/kmer_count_bench.py 2000000 100000 > gpurun_out/kmer_count.txt 2>&1; tail -2 gpurun_out/kmer_count.txt
T4_TIMING=1 python tools/e2e_cells_time.py 100000 1000 4 8 2>&1 | grep "pairs 100000\|timing" > gpurun_out/e2e_host.txt; head -1 gpurun_out/e2e_host.txt
T4_GPU_KMERCOUNT=1 T4_GPU_MATEOVERLAP=1 T4_TIMING=1 python tools/e2e_cells_time.py 100000 1000 4 8 2>&1 | grep "pairs 100000\|timing" > gpurun_out/e2e_device.txt; head -1 gpurun_out/e2e_device.txt
echo "opt-in legs done $(( $(date +%s) - T0 )) s"
