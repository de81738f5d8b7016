#!/bin/bash
# round 3, session a: GPU tests of HEAD, per-round / per-read latency log at 100 k pairs, config C2 itself (1 M pairs) with md5 sums
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
W=/tmp/w3a; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
( time T4_TIMING=1 T4_ROUND_LOG=$O/rounds_100k.txt T4_STATS_JSON=$O/stats_100k.json timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/m100 ) > $O/log_100k.txt 2>&1
md5sum $W/m100_raw.out $W/m100_assembled_reads.fa $W/m100_final.out >> $O/log_100k.txt
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
md5sum $W/c2_1.fq $W/c2_2.fq > $O/c2_md5.txt
( time T4_TIMING=1 T4_STATS_JSON=$O/stats_c2.json timeout 900 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/mc2 ) > $O/log_c2.txt 2>&1
md5sum $W/mc2_raw.out $W/mc2_assembled_reads.fa $W/mc2_final.out >> $O/c2_md5.txt
tail -5 $O/gpu_tests.txt; tail -8 $O/log_100k.txt; tail -6 $O/log_c2.txt; cat $O/c2_md5.txt
