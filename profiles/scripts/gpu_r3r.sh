#!/bin/bash
# round 3, session r: config C2 with run-trust4's DEFAULT options (mate-pair extension tail included) through oracle/_ref/trust4-dropin
# = the reference's main.cpp bound to libt4hip.so; expected md5 of _final.out: tests/golden/c2_digests.json, modes.default
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3r; mkdir -p $O
W=/tmp/w3r; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
md5sum $W/c2_1.fq $W/c2_2.fq > $O/c2_dropin_md5.txt
( time env T4_TIMING=1 timeout 680 oracle/_ref/trust4-dropin -t 8 -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/dr ) > $O/log_c2_dropin.txt 2>&1
echo "rc $?" >> $O/log_c2_dropin.txt
md5sum $W/dr_raw.out $W/dr_assembled_reads.fa $W/dr_final.out >> $O/c2_dropin_md5.txt
grep -v "Read in and count\|Processed [0-9]* reads" $O/log_c2_dropin.txt | tail -25 | cut -c1-300; cat $O/c2_dropin_md5.txt
