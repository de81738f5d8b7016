#!/bin/bash
# round 3, session h (as g, after the border-quirk fix and the eight-per-wavefront lean extension): GPU suite, bulk 100 k timing (path-state extension, LDS-code seed replay), phase profile, C2
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3h; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
tail -4 $O/gpu_tests.txt
W=/tmp/w3h; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
for i in 1 2; do
( time env T4_TIMING=1 T4_ROUND_LOG=$O/rounds_$i.txt timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/m100 ) > $O/log_100k_$i.txt 2>&1
md5sum $W/m100_raw.out >> $O/log_100k_$i.txt
grep "real\|raw.out\|first launch" $O/log_100k_$i.txt
done
( time env T4_TIMING=1 T4_PHASE_DUMP=1 LD_LIBRARY_PATH=$PWD/trust4_amd/variants/phases timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/mph ) > $O/log_100k_phases.txt 2>&1
md5sum $W/mph_raw.out >> $O/log_100k_phases.txt
grep "phase .* lds\|debug counters\|real\|raw.out" $O/log_100k_phases.txt | tail -30
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
( time T4_TIMING=1 T4_STATS_JSON=$O/stats_c2.json timeout 900 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/mc2 ) > $O/log_c2.txt 2>&1
md5sum $W/mc2_raw.out $W/mc2_assembled_reads.fa $W/mc2_final.out > $O/c2_md5.txt
grep "real" $O/log_c2.txt; cat $O/c2_md5.txt
