#!/bin/bash
# round 6, session s: bench.py as the driver runs it at N = 1 (20 steps + 5 warm-up of config C2, every leg in its place, inside 1800 s)
# gpurun --timeout 2100 -- 'bash profiles/scripts/gpu_r6s.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6s; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_flow.json 2> $O/bench_driver_flow.err; echo "bench rc $?"; grep real $O/bench_driver_flow.err
python3 - $O/bench_driver_flow.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step")}, d["config"]["is_baseline_config_c2"], d.get("parity_on_bench_batch"), d["c2"]["workload_decision"])
print("cells_1m", {k: d.get("stage1_cells_1m", {}).get(k) for k in ("seconds", "identical")}, "side legs skipped:", d.get("side_legs_skipped"), "keys", [k for k in d.keys()])
PY
