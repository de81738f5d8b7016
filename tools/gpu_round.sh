set -x
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_r01b.json 2> gpurun_out/bench_r01b.err; tail -c 3000 gpurun_out/bench_r01b.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_b -o r01b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e-pairs 0 > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/prof_b | head
find $GRAFT_REPO_ROOT/gpurun_out/prof_b -name "*kernel_stats*" | head -2 | xargs -I{} head -12 {}
