#!/bin/bash
# round 4, session s: barcode mode at 1 M pairs / 10 k cells (C5 recipe, seed 4: the sample of profiles/r03t_*) with the cells in
# 1 / 2 / 4 / 8 groups (T4_CELL_GROUPS), -t 32; outputs against round 3's digests (57cc18cd..., 89b90b07...)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s; mkdir -p $O
W=/tmp/w4s; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 0 4 $W/c5 --cells 10000 > /dev/null
echo "synth done $SECONDS"
for g in 1 4 2 8; do
  ( time env T4_TIMING=1 T4_CELL_GROUPS=$g T4_STATS_JSON=$O/stats_c5_g$g.json timeout 100 trust4_amd/bin/trust4-hip -t 32 -f $W/ref.fa -1 $W/c5_1.fq -2 $W/c5_2.fq --barcode $W/c5_bc.fa --UMI $W/c5_umi.fa -o $W/c5o ) > $O/log_c5_g$g.txt 2>&1
  md5sum $W/c5o_raw.out $W/c5o_assembled_reads.fa | cut -c1-32 | tr '\n' ' ' >> $O/log_c5_g$g.txt
  echo "== groups $g: $(grep real $O/log_c5_g$g.txt) $(grep 'Assembly rounds' $O/log_c5_g$g.txt | cut -c28-) $(tail -1 $O/log_c5_g$g.txt)"
  echo "elapsed $SECONDS"
  if [ $SECONDS -gt 130 ]; then break; fi
done
