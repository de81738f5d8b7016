#!/bin/bash
# round 4, session d: query lanes with the two classes of work (restricted re-queries / whole queries), 100 k pairs A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4d; mkdir -p $O
W=/tmp/w4d; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
run() {
  local name=$1 pre=$2 lim=$3; shift; shift; shift
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$name.json "$@" timeout $lim trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 ${pre}_1.fq -2 ${pre}_2.fq -o $W/m_$name ) > $O/log_$name.txt 2>&1
  md5sum $W/m_${name}_raw.out $W/m_${name}_assembled_reads.fa >> $O/log_$name.txt
  echo "== $name: $(grep -h 'real' $O/log_$name.txt | tr '\n' ' ') $(grep -o 'GPU query rounds [0-9]* with [0-9]* reads' $O/log_$name.txt) $(grep -o 'restricted re-queries: [0-9]* entries' $O/log_$name.txt) $(grep -o 'query lanes.*' $O/log_$name.txt | cut -c1-170)"
  grep -o '"kernel_ms": [0-9.]*' $O/stats_$name.json | tail -1
  tail -2 $O/log_$name.txt | cut -c1-34
}
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
run lanes1 $W/b 300
run lanes2 $W/b 300 T4_LIVE_LANES=2
run lanes3 $W/b 300 T4_LIVE_LANES=3
run lanes2_nowide $W/b 300 T4_LIVE_LANES=2 T4_WIDE_OFF=1
run lanes1_nowide $W/b 300 T4_WIDE_OFF=1
run lanes2_ahead $W/b 300 T4_LIVE_LANES=2 T4_QUERY_AHEAD=64
