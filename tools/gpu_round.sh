# one GPU session: smoke, bench (JSON kept), rocprofv3 kernel stats of the same command, PMC passes. Outputs under gpurun_out/.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -o pmc -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --e2e-pairs 0 > /dev/null 2>&1
done
F=$(find $R/gpurun_out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $R/gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
cp $F $R/gpurun_out/pmc_FETCH_SIZE.csv; cp $W $R/gpurun_out/pmc_WRITE_SIZE.csv
python $R/tools/pmc_summary.py $F $W $R/profiles/r01_pmc_summary.json
cp $R/profiles/r01_pmc_summary.json $R/gpurun_out/pmc_summary.json
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r01 -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e-pairs 0 > /dev/null 2>&1
find $R/gpurun_out/prof -name "*kernel_stats*" | head -1 | xargs -I{} cp {} $R/gpurun_out/kernel_stats.csv
head -9 $R/gpurun_out/kernel_stats.csv
cd $R && python bench.py --steps 3 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.json
