# one GPU session: smoke, bench (JSON kept), rocprofv3 kernel stats of the same command (CSV kept)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
python bench.py --steps 3 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.json | cut -c1-400
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r01 -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e-pairs 0 > /dev/null 2>&1
find $R/gpurun_out/prof -name "*kernel_stats*" | head -1 | xargs -I{} head -14 {}
