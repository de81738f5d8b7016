#!/bin/bash
# round 4, session u: what the allocator does to the input and count phases of barcode mode at 1 M pairs (-t 32): records moved through
# ProcessRead (default) vs copied (T4_PR_COPY), with and without the mallopt settings
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4u; mkdir -p $O
W=/tmp/w4u; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 0 4 $W/c5 --cells 10000 > /dev/null
run() { tag=$1; shift
  ( time env T4_TIMING=1 "$@" T4_STATS_JSON=$O/stats_$tag.json timeout 60 trust4_amd/bin/trust4-hip -t 32 -f $W/ref.fa -1 $W/c5_1.fq -2 $W/c5_2.fq --barcode $W/c5_bc.fa --UMI $W/c5_umi.fa -o $W/c5o ) > $O/log_$tag.txt 2>&1
  md5sum $W/c5o_raw.out $W/c5o_assembled_reads.fa | cut -c1-32 | tr '\n' ' ' >> $O/log_$tag.txt
  echo "== $tag: $(grep -h 'real\|^sys' $O/log_$tag.txt | tr '\n' ' ') $(tail -1 $O/log_$tag.txt)"; grep -h "timing: input\|timing: 21\|timing: count" $O/log_$tag.txt | cut -c28-150; python3 -c "import json;print(json.load(open('$O/stats_$tag.json'))['phases_s'])"; }
run move_mallopt T4_X=1
run copy_mallopt T4_PR_COPY=1
run move_plain T4_NO_MALLOPT=1
echo "elapsed $SECONDS"
