#!/bin/bash
# round 3, session t (last): HEAD after the host-side input work: the whole GPU suite, the bench batch and C5 (1 M pairs, seed 4) timed, a short bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3t; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
tail -3 $O/gpu_tests.txt; echo "elapsed $SECONDS"
W=/tmp/w3t; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 0 4 $W/c5 --cells 10000 > /dev/null
for i in 1 2; do ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c5_t32_$i.json timeout 200 trust4_amd/bin/trust4-hip -t 32 -f $W/ref.fa -1 $W/c5_1.fq -2 $W/c5_2.fq --barcode $W/c5_bc.fa --UMI $W/c5_umi.fa -o $W/c5o ) > $O/log_c5_t32_$i.txt 2>&1
md5sum $W/c5o_raw.out $W/c5o_assembled_reads.fa >> $O/log_c5_t32_$i.txt; grep "real\|input loop\|c5o_" $O/log_c5_t32_$i.txt | cut -c1-160; done
echo "elapsed $SECONDS"
if [ $SECONDS -lt 330 ]; then timeout 170 python bench.py --c2 0 --traffic 0 --side-legs 0 > $O/bench_short.json 2> $O/bench_short.err; echo "bench rc $?"; cut -c1-330 $O/bench_short.json; fi
