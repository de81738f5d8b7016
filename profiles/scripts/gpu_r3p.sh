#!/bin/bash
# round 3, session p: window entries whose group statistics cannot move keep their result across tolerated index edits (exact rule from the query itself): tests, 100 k A/B, C2
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3p; mkdir -p $O
timeout 600 python -m pytest tests/test_stage1_e2e.py -m gpu -q -k "stable_group or window_validity" > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
tail -4 $O/gpu_tests.txt
W=/tmp/w3p; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
run() {
  local name=$1 pre=$2; shift; shift
  ( time env T4_TIMING=1 "$@" timeout 600 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 ${pre}_1.fq -2 ${pre}_2.fq -o $W/m_$name ) > $O/log_$name.txt 2>&1
  md5sum $W/m_${name}_raw.out $W/m_${name}_assembled_reads.fa >> $O/log_$name.txt
  echo "== $name: $(grep -h 'real' $O/log_$name.txt | tr '\n' ' ') $(grep -o 'first launch to sync [0-9.]*' $O/log_$name.txt) $(grep -o 'GPU query rounds [0-9]* with [0-9]* reads' $O/log_$name.txt) $(grep -o 'tolerated index edits.*' $O/log_$name.txt | cut -c1-200)"
  tail -2 $O/log_$name.txt | cut -c1-34
}
run stable $W/b
run budget $W/b T4_NO_STABLE_STATS=1
run stable2 $W/b
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
run c2 $W/c2
