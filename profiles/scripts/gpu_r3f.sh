#!/bin/bash
# round 3, session f: tandem-repeat skip-rule case with both seed replays; bulk 100 k with lean extension records
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3f; mkdir -p $O
python tools/ab_skiprule.py > $O/ab_new.txt 2>&1
T4_LIB=$PWD/trust4_amd/variants/seedserial/libt4hip.so python tools/ab_skiprule.py > $O/ab_serial.txt 2>&1
tail -4 $O/ab_new.txt; tail -4 $O/ab_serial.txt
W=/tmp/w3f; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
for i in 1 2; do
( time env T4_TIMING=1 timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/m100 ) > $O/log_100k_$i.txt 2>&1
md5sum $W/m100_raw.out >> $O/log_100k_$i.txt
grep "real\|raw.out\|first launch" $O/log_100k_$i.txt
done
