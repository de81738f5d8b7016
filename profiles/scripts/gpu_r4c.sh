#!/bin/bash
# round 4, session c: restricted re-queries + the scatter fix on the hardware: tests (wide, window validity under T4_VERIFY_WINDOW), 100 k pairs A/B, C2
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c; mkdir -p $O
timeout 900 python -m pytest tests/test_wide_query.py tests/test_stage1_e2e.py -m gpu -q -x -k "wide or window_validity_rules_gpu" > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
tail -4 $O/gpu_tests.txt
W=/tmp/w4c; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
run() {
  local name=$1 pre=$2 lim=$3; shift; shift; shift
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$name.json "$@" timeout $lim trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 ${pre}_1.fq -2 ${pre}_2.fq -o $W/m_$name ) > $O/log_$name.txt 2>&1
  md5sum $W/m_${name}_raw.out $W/m_${name}_assembled_reads.fa >> $O/log_$name.txt
  echo "== $name: $(grep -h 'real' $O/log_$name.txt | tr '\n' ' ') $(grep -o 'first launch to sync [0-9.]*' $O/log_$name.txt) $(grep -o 'GPU query rounds [0-9]* with [0-9]* reads' $O/log_$name.txt) $(grep -o 'restricted re-queries.*' $O/log_$name.txt | cut -c1-150)"
  grep -o '"kernel_ms": [0-9.]*' $O/stats_$name.json | tail -1
  tail -2 $O/log_$name.txt | cut -c1-34
}
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
run 100k $W/b 300
run 100k_norestrict $W/b 300 T4_RESTRICT_OFF=1
run 100k_norestrict_nowide $W/b 300 T4_RESTRICT_OFF=1 T4_WIDE_OFF=1
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
run c2 $W/c2 600
