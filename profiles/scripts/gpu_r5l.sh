#!/bin/bash
# round 5, session l: bench.py's N = 1 flow on a GPU with the 20 k-pair stand-in for C2 (T4_BENCH_C2_STANDIN=c2mini: the run of "C2",
# the reference's prefix timing before the steps, the choice of the workload, the PMC passes) and the kernel-parity part of the GPU suite
# through HEAD (the run sizes of the candidate records changed chainFinish / the scoring passes).
# gpurun --timeout 300 -- 'bash profiles/scripts/gpu_r5l.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5l; mkdir -p $O
export TMPDIR=/tmp
T4_BENCH_C2_STANDIN=c2mini timeout 170 python bench.py --steps 3 --warmup 2 --side-legs 0 --cpu-c2-pairs 10000 > $O/bench_standin.json 2> $O/bench_standin.err; echo "bench rc $?"; tail -c 400 $O/bench_standin.err
python3 - $O/bench_standin.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print({k: d[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "scaling")})
print(d["config"]["workload"][:160], d["config"]["is_baseline_config_c2"])
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "traffic", "kernel_ms", "traffic_over_algorithmic")})
print("cpu_baseline", d.get("cpu_baseline"))
print("decision", d["c2"].get("workload_decision"), "c2 seconds", d["c2"].get("seconds"))
PY
echo "elapsed $SECONDS"
timeout 170 python -m pytest tests/test_gpu_parity.py tests/test_wide_query.py tests/test_lean_records.py tests/test_zz_kmer_count_gpu.py -m gpu -q -x > $O/gpu_tests_kernels.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_kernels.txt; tail -3 $O/gpu_tests_kernels.txt | cut -c1-300
echo "elapsed $SECONDS"
