#!/bin/bash
# round 4, session l (final): the whole GPU suite, bench.py as the driver runs it (--steps 20 --warmup 5), the C2-is-the-workload branch, the first 500 k pairs of C3 as a config leg
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4l; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt; tail -5 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt | cut -c1-300
( time T4_BENCH_PMC_DIR=$O python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_invocation.json 2> $O/bench_driver.err ) 2> $O/bench_driver_time.txt; tail -c 300 $O/bench_driver.err; cat $O/bench_driver_time.txt
( time python bench.py --steps 1 --warmup 1 --side-legs 1 --traffic 0 --cpu-c2-pairs 200000 --config-leg c3p05 > $O/bench_c2_workload.json 2> $O/bench_c2.err ) 2> $O/bench_c2_time.txt; tail -c 300 $O/bench_c2.err; cat $O/bench_c2_time.txt
python3 - <<PY
import json
for name in ("bench_driver_invocation", "bench_c2_workload"):
    try:
        d = json.load(open("$O/%s.json" % name))
    except Exception as e:
        print(name, "unreadable", e); continue
    print("==", name, {k: d[k] for k in ("value", "ms_per_step", "steps", "warmup")}, d["config"]["workload"][:110])
    r = d["roofline"]; print("   roofline", {k: r.get(k) for k in ("achieved", "frac", "traffic", "traffic_over_algorithmic", "kernel_ms", "launch_ms_avg")})
    print("   cpu_baseline", d.get("cpu_baseline"), "parity", d.get("parity_on_bench_batch"))
    c2 = d.get("c2") or {}; print("   c2", {k: c2.get(k) for k in ("seconds", "pairs_per_s", "identical", "rounds", "reads_queried", "reads_served", "kernel_ms", "addread_pass_s", "error")}, (c2.get("cpu_baseline") or {}).get("value"))
    c3 = d.get("c3p05") or {}; print("   c3p05", {k: c3.get(k) for k in ("seconds", "pairs_per_s", "identical", "rounds", "reads_queried", "reads_served", "kernel_ms", "contigs", "error")})
    print("   cells", (d.get("stage1_cells") or {}).get("pairs_per_s"), (d.get("stage1_cells") or {}).get("identical"), "stage0", (d.get("stage0_e2e") or {}).get("pairs_per_s"), "annot", (d["passes"].get("rough_annotation_c2") or {}).get("kernel_ms"))
PY
