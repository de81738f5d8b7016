#!/bin/bash
# round 4, session e: two lanes, larger background batches
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4e; mkdir -p $O
W=/tmp/w4e; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
run() {
  local name=$1 pre=$2 lim=$3; shift; shift; shift
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$name.json "$@" timeout $lim trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 ${pre}_1.fq -2 ${pre}_2.fq -o $W/m_$name ) > $O/log_$name.txt 2>&1
  md5sum $W/m_${name}_raw.out $W/m_${name}_assembled_reads.fa >> $O/log_$name.txt
  echo "== $name: $(grep -h 'real' $O/log_$name.txt | tr '\n' ' ') $(grep -o 'GPU query rounds [0-9]* with [0-9]* reads' $O/log_$name.txt) $(grep -o 'query lanes.*' $O/log_$name.txt | cut -c1-170)"
  grep -o 'assembler host seconds.*' $O/log_$name.txt | cut -c1-300
  grep -o '"kernel_ms": [0-9.]*' $O/stats_$name.json | tail -1
  tail -2 $O/log_$name.txt | cut -c1-34
}
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
run l1 $W/b 300 T4_WIDE_OFF=1
run l2_b32_a64 $W/b 300 T4_WIDE_OFF=1 T4_LIVE_LANES=2 T4_LIVE_MIN_BATCH=32 T4_QUERY_AHEAD=64
run l2_b48_a128 $W/b 300 T4_WIDE_OFF=1 T4_LIVE_LANES=2 T4_LIVE_MIN_BATCH=48 T4_QUERY_AHEAD=128
run l2_b16_a48 $W/b 300 T4_WIDE_OFF=1 T4_LIVE_LANES=2 T4_LIVE_MIN_BATCH=16 T4_QUERY_AHEAD=48
run l1_a8 $W/b 300 T4_WIDE_OFF=1 T4_QUERY_AHEAD=8
run l1_a64 $W/b 300 T4_WIDE_OFF=1 T4_QUERY_AHEAD=64
