#!/bin/bash
# round 4, session a: the wide AddRead query (csrc/t4_wide.h) on the hardware for the first time -- its tests, the bulk paths, then
# 100 k pairs / C2 / the first 500 k pairs of C3 with timing lines, and the reference on the same box for C2's first 200 k pairs
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests/test_wide_query.py tests/test_stage1_e2e.py -m gpu -q -x -k "wide or bulk_live_set_paths" > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
tail -6 $O/gpu_tests.txt
W=/tmp/w4a; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
run() {
  local name=$1 pre=$2 lim=$3; shift; shift; shift
  ( time env T4_TIMING=1 "$@" timeout $lim trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 ${pre}_1.fq -2 ${pre}_2.fq -o $W/m_$name ) > $O/log_$name.txt 2>&1
  md5sum $W/m_${name}_raw.out $W/m_${name}_assembled_reads.fa >> $O/log_$name.txt
  echo "== $name: $(grep -h 'real' $O/log_$name.txt | tr '\n' ' ') $(grep -o 'first launch to sync [0-9.]*' $O/log_$name.txt) $(grep -o 'GPU query rounds [0-9]* with [0-9]* reads' $O/log_$name.txt) $(grep -o 'wide query served.*' $O/log_$name.txt | cut -c1-160)"
  tail -2 $O/log_$name.txt | cut -c1-34
}
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
run 100k $W/b 300
run 100k_wideoff $W/b 300 T4_WIDE_OFF=1
( time oracle/_ref/trust4 -t $(nproc) --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/r_100k ) > $O/ref_100k.txt 2>&1; md5sum $W/r_100k_raw.out $W/r_100k_assembled_reads.fa >> $O/ref_100k.txt; tail -5 $O/ref_100k.txt | cut -c1-60
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
run c2 $W/c2 600
head -800000 $W/c2_1.fq > $W/c2h_1.fq; head -800000 $W/c2_2.fq > $W/c2h_2.fq
( time oracle/_ref/trust4 -t $(nproc) --skipMateExtension -f $W/ref.fa -1 $W/c2h_1.fq -2 $W/c2h_2.fq -o $W/r_c2h ) > $O/ref_c2_first200k.txt 2>&1; grep real $O/ref_c2_first200k.txt
tools/t4synth $W/ref.fa 500000 200000 2 $W/c3 > /dev/null
md5sum $W/c3_1.fq $W/c3_2.fq > $O/c3p05_inputs_md5.txt
run c3p05 $W/c3 900
nproc; lscpu | grep "Model name"
