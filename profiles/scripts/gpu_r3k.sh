#!/bin/bash
# round 3, session k: RCCL gather test (one rank), seed/skip-rule parity, then the whole default bench line + a C3 prefix leg
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3k; mkdir -p $O
timeout 600 python -m pytest tests/test_stage1_e2e.py tests/test_gpu_parity.py -m gpu -q -k "rccl or novel_min or bulk_live" > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
tail -3 $O/gpu_tests.txt
( time timeout 2400 python bench.py --config-leg c3p2 > $O/bench.json 2> $O/bench.err ); tail -c 400 $O/bench.err
python - <<'PY'
import json
b = json.load(open("gpurun_out/r3k/bench.json"))
print("value", b["value"], "ms/step", b["ms_per_step"], "parity", b.get("parity_on_bench_batch"))
print("roofline", {k: b["roofline"][k] for k in ("achieved", "frac", "traffic", "kernel_ms", "launch_ms_avg") if k in b["roofline"]}, b["roofline"].get("traffic_detail", {}).get("error"))
for k in ("c2", "c3p2", "stage1_cells", "stage0_e2e"):
    print(k, json.dumps(b.get(k))[:600])
print("cpu", json.dumps(b.get("cpu_baseline"))[:300])
PY
