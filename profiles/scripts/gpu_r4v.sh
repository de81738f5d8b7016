#!/bin/bash
# round 4, session v (last): barcode mode at 1 M pairs / 10 k cells (-t 32) at HEAD -- reader blocks recycled, output formatted on the
# threads, trimming loops on the threads -- with the default 4 096 cells in flight and with 16 384
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4v; mkdir -p $O
W=/tmp/w4v; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 0 4 $W/c5 --cells 10000 > /dev/null
run() { tag=$1; shift
  ( time env T4_TIMING=1 "$@" T4_STATS_JSON=$O/stats_$tag.json timeout 40 trust4_amd/bin/trust4-hip -t 32 -f $W/ref.fa -1 $W/c5_1.fq -2 $W/c5_2.fq --barcode $W/c5_bc.fa --UMI $W/c5_umi.fa -o $W/c5o ) > $O/log_$tag.txt 2>&1
  md5sum $W/c5o_raw.out $W/c5o_assembled_reads.fa $W/c5o_final.out | cut -c1-32 | tr '\n' ' ' >> $O/log_$tag.txt
  echo "== $tag: $(grep -h 'real\|^sys' $O/log_$tag.txt | tr '\n' ' ') $(tail -1 $O/log_$tag.txt)"; grep -h "timing: input\|timing: 21\|timing: count\|timing: waited\|Cell groups\|Assembly rounds" $O/log_$tag.txt | cut -c28-190; python3 -c "import json;print(json.load(open('$O/stats_$tag.json'))['phases_s'])"; }
run head T4_X=1
run lanes16k T4_LANES=16384
echo "elapsed $SECONDS"
