#!/bin/bash
# round 3, session u: reads whose last query was slow stay out of launches made for reads well before them (T4_SLOW_US / T4_SLOW_AHEAD): sweep on the bench batch, C2 with the best setting
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3u; mkdir -p $O
W=/tmp/w3u; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null &
best=""; bestt=999999
for cfg in "0 3" "300 3" "500 3" "300 6"; do
  set -- $cfg
  s=$(date +%s%N)
  env T4_TIMING=1 T4_SLOW_US=$1 T4_SLOW_AHEAD=$2 timeout 60 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/m > $O/log_$1_$2.txt 2>&1
  e=$(date +%s%N); ms=$(( (e - s) / 1000000 ))
  echo "== slow_us $1 ahead $2: $ms ms  $(md5sum $W/m_raw.out | cut -c1-12) $(grep -o 'GPU query rounds [0-9]* with [0-9]* reads' $O/log_$1_$2.txt) $(grep -o 'first launch to sync [0-9.]*' $O/log_$1_$2.txt)" | tee -a $O/sweep.txt
  if [ "$1" != "0" ] && [ $ms -lt $bestt ]; then bestt=$ms; best="$cfg"; fi
done
wait
set -- $best
echo "C2 with slow_us $1 ahead $2 (elapsed $SECONDS s)" | tee -a $O/sweep.txt
if [ $SECONDS -lt 75 ]; then
  ( time env T4_TIMING=1 T4_SLOW_US=$1 T4_SLOW_AHEAD=$2 timeout 125 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/c2o ) > $O/log_c2.txt 2>&1
  md5sum $W/c2o_raw.out $W/c2o_assembled_reads.fa >> $O/log_c2.txt; grep "real\|GPU query rounds" $O/log_c2.txt | cut -c1-200; tail -2 $O/log_c2.txt | cut -c1-34
fi
