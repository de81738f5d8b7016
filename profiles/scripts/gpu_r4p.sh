#!/bin/bash
# round 4, session p: C2 with 16 and 32 host threads
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4p; mkdir -p $O
W=/tmp/w4p; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
for t in 16 32; do
  ( time env T4_TIMING=1 trust4_amd/bin/trust4-hip -t $t --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/m_$t ) > $O/log_c2_t$t.txt 2>&1
  md5sum $W/m_${t}_raw.out >> $O/log_c2_t$t.txt
  echo "== -t $t: $(grep real $O/log_c2_t$t.txt) $(grep -o 'assembler host seconds.*' $O/log_c2_t$t.txt | cut -c1-260)"; tail -1 $O/log_c2_t$t.txt | cut -c1-34
done
