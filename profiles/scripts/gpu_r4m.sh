#!/bin/bash
# round 4, session m: partitions by the quantiles of a sample of the hits' contigs: wide tests, the first 500 k pairs of C3, C2
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4m; mkdir -p $O
timeout 600 python -m pytest tests/test_wide_query.py tests/test_stage1_e2e.py -m gpu -q -x -k "wide or bulk_live_set_paths" > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt
W=/tmp/w4m; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
run() {
  local name=$1 pre=$2 lim=$3; shift; shift; shift
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$name.json "$@" timeout $lim trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 ${pre}_1.fq -2 ${pre}_2.fq -o $W/m_$name ) > $O/log_$name.txt 2>&1
  md5sum $W/m_${name}_raw.out $W/m_${name}_assembled_reads.fa >> $O/log_$name.txt
  echo "== $name: $(grep -h 'real' $O/log_$name.txt | tr '\n' ' ') $(grep -o 'GPU query rounds [0-9]* with [0-9]* reads' $O/log_$name.txt) $(grep -o 'wide query served [0-9]* window entries' $O/log_$name.txt)"
  grep -o 'wide query: .*' $O/log_$name.txt | cut -c1-300
  grep -o 'entries that fell whole.*' $O/log_$name.txt | cut -c1-400
  grep -o 'restricted re-queries.*' $O/log_$name.txt | cut -c1-200
  grep -o '"kernel_ms": [0-9.]*' $O/stats_$name.json | tail -1
  tail -2 $O/log_$name.txt | cut -c1-34
}
tools/t4synth $W/ref.fa 500000 200000 2 $W/c3 > /dev/null
run c3p05 $W/c3 600
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
run c2 $W/c2 600
