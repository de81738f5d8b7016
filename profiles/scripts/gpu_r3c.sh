#!/bin/bash
# round 3, session c: bulk 100 k pairs with the traceback skip (default lanes), per-phase profile (phases variant), barcode mode at 1 M pairs / 10 k cells
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c; mkdir -p $O
W=/tmp/w3c; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
for i in 1 2; do
( time env T4_TIMING=1 timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/m100 ) > $O/log_100k_$i.txt 2>&1
md5sum $W/m100_raw.out >> $O/log_100k_$i.txt
done
( time env T4_TIMING=1 T4_PHASE_DUMP=1 LD_LIBRARY_PATH=$PWD/trust4_amd/variants/phases timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/mph ) > $O/log_100k_phases.txt 2>&1
md5sum $W/mph_raw.out >> $O/log_100k_phases.txt
# barcode mode
tools/t4synth $W/ref.fa 1000000 0 4 $W/c5 --cells 10000 > /dev/null
A="-f $W/ref.fa -1 $W/c5_1.fq -2 $W/c5_2.fq --barcode $W/c5_bc.fa --UMI $W/c5_umi.fa"
for t in 8 32; do
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c5_t$t.json timeout 600 trust4_amd/bin/trust4-hip -t $t $A -o $W/c5m$t ) > $O/log_c5_t$t.txt 2>&1
md5sum $W/c5m${t}_raw.out $W/c5m${t}_final.out $W/c5m${t}_assembled_reads.fa >> $O/log_c5_t$t.txt
done
( time timeout 900 oracle/_ref/trust4 -t 32 $A -o $W/c5ref ) > $O/log_c5_ref_t32.txt 2>&1
md5sum $W/c5ref_raw.out $W/c5ref_final.out $W/c5ref_assembled_reads.fa >> $O/log_c5_ref_t32.txt
grep -h "real\|raw.out" $O/log_100k_1.txt $O/log_100k_2.txt $O/log_c5_t8.txt $O/log_c5_t32.txt $O/log_c5_ref_t32.txt
grep "phase " $O/log_100k_phases.txt | tail -45
