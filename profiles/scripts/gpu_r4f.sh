#!/bin/bash
# round 4, session f: the new bench.py end to end (default invocation) + C2 with the wide query from 32768 hits / from 8192 hits / off
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4f; mkdir -p $O
( time T4_BENCH_PMC_DIR=$O python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; tail -c 600 $O/bench.err; cat $O/bench_time.txt
python3 - <<PY
import json
d=json.load(open("$O/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","steps","warmup")}, d["config"]["workload"][:160])
print("roofline", {k:d["roofline"][k] for k in ("achieved","frac","traffic","kernel_ms","launch_ms_avg") if k in d["roofline"]}, d["roofline"].get("traffic_over_algorithmic"))
print("cpu_baseline", d.get("cpu_baseline"))
c2=d.get("c2",{}); print("c2", {k:c2.get(k) for k in ("seconds","pairs_per_s","identical","rounds","reads_queried","reads_served","kernel_ms","addread_pass_s","cpu_baseline","error")})
print("cells", d.get("stage1_cells")); print("stage0", d.get("stage0_e2e")); print("annot", {k:d["passes"].get("rough_annotation_c2",{}).get(k) for k in ("kernel_ms","reads_per_s","error")})
PY
W=/tmp/w4f; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
run() {
  local name=$1 pre=$2 lim=$3; shift; shift; shift
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$name.json "$@" timeout $lim trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 ${pre}_1.fq -2 ${pre}_2.fq -o $W/m_$name ) > $O/log_$name.txt 2>&1
  md5sum $W/m_${name}_raw.out $W/m_${name}_assembled_reads.fa >> $O/log_$name.txt
  echo "== $name: $(grep -h 'real' $O/log_$name.txt | tr '\n' ' ') $(grep -o 'GPU query rounds [0-9]* with [0-9]* reads' $O/log_$name.txt) $(grep -o 'restricted re-queries: [0-9]* entries' $O/log_$name.txt) $(grep -o 'wide query served [0-9]* window entries' $O/log_$name.txt)"
  grep -o 'assembler host seconds.*' $O/log_$name.txt | cut -c1-300
  grep -o '"kernel_ms": [0-9.]*' $O/stats_$name.json | tail -1
  tail -2 $O/log_$name.txt | cut -c1-34
}
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
run c2_wideoff $W/c2 600 T4_WIDE_OFF=1
run c2_wide8k $W/c2 600 T4_WIDE_MIN_HITS=8192
