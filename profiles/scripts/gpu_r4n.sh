#!/bin/bash
# round 4, session n: the load factor of the partitions is kept across calls only after a key overflow: C2 and the first 500 k pairs of C3 again
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4n; mkdir -p $O
W=/tmp/w4n; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
run() {
  local name=$1 pre=$2 lim=$3; shift; shift; shift
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$name.json "$@" timeout $lim trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 ${pre}_1.fq -2 ${pre}_2.fq -o $W/m_$name ) > $O/log_$name.txt 2>&1
  md5sum $W/m_${name}_raw.out $W/m_${name}_assembled_reads.fa >> $O/log_$name.txt
  echo "== $name: $(grep -h 'real' $O/log_$name.txt | tr '\n' ' ') $(grep -o 'GPU query rounds [0-9]* with [0-9]* reads' $O/log_$name.txt)"
  grep -o 'wide query: .*' $O/log_$name.txt | cut -c1-300
  grep -o 'assembler host seconds.*' $O/log_$name.txt | cut -c1-300
  grep -o '"kernel_ms": [0-9.]*' $O/stats_$name.json | tail -1
  tail -2 $O/log_$name.txt | cut -c1-34
}
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
run c2 $W/c2 600
tools/t4synth $W/ref.fa 500000 200000 2 $W/c3 > /dev/null
run c3p05 $W/c3 600
