#!/bin/bash
# round 3, session l: argument structs in LDS (scratch 736 -> 80 B/lane): parity subset, bulk timing + PMC traffic (bench legs),
# rough-annotation pass A/B against the round-2 library, barcode mode 1 M pairs with the host-side changes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_lean_records.py tests/test_stage1_e2e.py tests/test_stage0_e2e.py -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
tail -3 $O/gpu_tests.txt
for v in "" r2final; do
  if [ -n "$v" ]; then export T4_LIB=$PWD/trust4_amd/variants/$v/libt4hip.so; else unset T4_LIB; fi
  python tools/gpu_pass.py 2000000 4 2>&1 | tail -1 | sed "s/^/annotate pass [${v:-head}]: /" | tee -a $O/annotate_ab.txt
done
unset T4_LIB
( time timeout 900 python bench.py --steps 2 --warmup 1 --c2 0 --side-legs 0 --cpu-single-pairs 0 > $O/bench_short.json 2> $O/bench_short.err ) 2>&1 | grep real
python - <<'PY'
import json
b = json.load(open("gpurun_out/r3l/bench_short.json"))
r = b["roofline"]
print("value", b["value"], "ms/step", b["ms_per_step"], "parity", b.get("parity_on_bench_batch"), "kernel_ms", r["kernel_ms"], "traffic", r["traffic"], "x alg", r.get("traffic_over_algorithmic"), r.get("traffic_detail", {}).get("raw"))
PY
W=/tmp/w3l; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 0 4 $W/c5 --cells 10000 > /dev/null
A="-f $W/ref.fa -1 $W/c5_1.fq -2 $W/c5_2.fq --barcode $W/c5_bc.fa --UMI $W/c5_umi.fa"
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c5_t32.json timeout 600 trust4_amd/bin/trust4-hip -t 32 $A -o $W/c5a ) > $O/log_c5_t32.txt 2>&1
md5sum $W/c5a_raw.out $W/c5a_assembled_reads.fa >> $O/log_c5_t32.txt
( time env T4_TIMING=1 T4_GPU_KMERCOUNT=1 T4_STATS_JSON=$O/stats_c5_t32_gpukc.json timeout 600 trust4_amd/bin/trust4-hip -t 32 $A -o $W/c5b ) > $O/log_c5_t32_gpukc.txt 2>&1
md5sum $W/c5b_raw.out $W/c5b_assembled_reads.fa >> $O/log_c5_t32_gpukc.txt
grep -h "real\|raw.out" $O/log_c5_t32.txt $O/log_c5_t32_gpukc.txt; cat $O/stats_c5_t32.json | head -c 400; echo; cat $O/stats_c5_t32_gpukc.json | head -c 300
