#!/bin/bash
# round 5, session g: the state of HEAD -- the whole GPU suite, smoke, and a complete bench line with C2 as the workload
# (--steps 2 --warmup 1: the driver's own invocation takes 25 C2 runs).
# gpurun --timeout 1500 -- 'bash profiles/scripts/gpu_r5g.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5g; mkdir -p $O
export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt; tail -4 $O/gpu_tests.txt | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt | cut -c1-300
echo "elapsed $SECONDS"
timeout 1000 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -c 600 $O/bench.err
python3 - $O/bench.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print({k: d[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "scaling")})
print(d["config"]["workload"][:120], d["config"]["is_baseline_config_c2"], d["config"]["phases_s"])
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "traffic", "kernel_ms", "frac_reads_served_only")})
print("chain", d.get("chain"))
print("cpu_baseline", d.get("cpu_baseline"))
c2 = d.get("c2") or {}
print("c2", {k: c2.get(k) for k in ("seconds", "pairs_per_s", "identical", "rounds", "reads_queried", "reads_served", "hits")}, (c2.get("cpu_baseline") or {}).get("value"))
for k in ("stage1_cells", "stage1_cells_1m", "stage0_e2e"):
    print(k, {x: (d.get(k) or {}).get(x) for x in ("seconds", "pairs_per_s", "identical", "phases_s")})
PY
echo "elapsed $SECONDS"
