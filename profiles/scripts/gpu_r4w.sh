#!/bin/bash
# round 4, session w: one run of barcode mode at 1 M pairs with the reader blocks NOT recycled (T4_NO_RECYCLE, a temporary switch): the input
# loop of session v's runs (1.29 s, ProcessRead 0.81 s) against session u's (0.95 s, 0.07 s) -- which of the two the recycling explains
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4w; mkdir -p $O
W=/tmp/w4w; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 0 4 $W/c5 --cells 10000 > /dev/null
( time env T4_TIMING=1 T4_NO_RECYCLE=1 T4_STATS_JSON=$O/stats.json timeout 30 trust4_amd/bin/trust4-hip -t 32 -f $W/ref.fa -1 $W/c5_1.fq -2 $W/c5_2.fq --barcode $W/c5_bc.fa --UMI $W/c5_umi.fa -o $W/c5o ) > $O/log.txt 2>&1
md5sum $W/c5o_raw.out $W/c5o_assembled_reads.fa | cut -c1-32 | tr '\n' ' ' >> $O/log.txt
grep -h "real\|^sys\|timing: input\|timing: 21\|timing: count" $O/log.txt | cut -c1-170; tail -1 $O/log.txt; cat $O/stats.json | cut -c1-260
