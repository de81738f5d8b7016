#!/bin/bash
# round 3, session s: block-wise input loop: e2e GPU tests, the bench batch, C5 at 1 M pairs (seed 4 as in sessions c / l: md5 57cc18cd...)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3s; mkdir -p $O
timeout 500 python -m pytest tests/test_stage1_e2e.py tests/test_run_trust4_dropin.py -m gpu -q -k "not window_validity and not stable_group" > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
tail -3 $O/gpu_tests.txt
W=/tmp/w3s; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
for i in 1 2; do ( time env T4_TIMING=1 timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/m ) > $O/log_100k_$i.txt 2>&1; md5sum $W/m_raw.out >> $O/log_100k_$i.txt; grep "real\|input loop\|raw.out" $O/log_100k_$i.txt | cut -c1-160; done
tools/t4synth $W/ref.fa 1000000 0 4 $W/c5 --cells 10000 > /dev/null
for i in 1 2; do ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c5_t32.json timeout 600 trust4_amd/bin/trust4-hip -t 32 -f $W/ref.fa -1 $W/c5_1.fq -2 $W/c5_2.fq --barcode $W/c5_bc.fa --UMI $W/c5_umi.fa -o $W/c5o ) > $O/log_c5_t32_$i.txt 2>&1
md5sum $W/c5o_raw.out $W/c5o_assembled_reads.fa >> $O/log_c5_t32_$i.txt; grep "real\|input loop\|c5o_" $O/log_c5_t32_$i.txt | cut -c1-160; done
