#!/bin/bash
# round 6, session x: barcode mode at 1 M pairs / 10 k cells -- cell groups, lanes and per-cell window again with this round's query call.
# gpurun --timeout 900 -- 'bash profiles/scripts/gpu_r6x.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6x; mkdir -p $O
export TMPDIR=/tmp
W=/tmp/w6x; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
tools/t4synth $W/ref.fa 1000000 0 4 $W/c5 --cells 10000 > /dev/null
run() {   # tag, env...
  tag=$1; shift
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$tag.json "$@" timeout 200 $BIN -t 32 -f $W/ref.fa -1 $W/c5_1.fq -2 $W/c5_2.fq --barcode $W/c5_bc.fa --UMI $W/c5_umi.fa -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  echo "$tag: $(md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_$tag.txt) $(python3 -c "import json;p=json.load(open('$O/stats_$tag.json'))['phases_s'];print('cell pass %.2f' % (p['assembled']-p['trimmed_ready']))") (57cc18cd 89b90b07 expected) elapsed $SECONDS"
  rm -f $W/o_${tag}_*
}
run warm
run g4
run g6 T4_CELL_GROUPS=6
run g8 T4_CELL_GROUPS=8
run g3 T4_CELL_GROUPS=3
run w6 T4_WINDOW=6
run w8 T4_WINDOW=8
run w3 T4_WINDOW=3
run l2k T4_LANES=2048
run l8k T4_LANES=8192
run g4b
