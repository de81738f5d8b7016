#!/bin/bash
# round 6, session f: HEAD (two-pass UpdateIndexFromRead with prefetch, window k-mers of the forward strand only) and a sweep of the
# round policy now that the host thread and the launch path are lighter: look-ahead, pending contigs, light rounds, the wide query's
# threshold, deferred extensions, host threads. Config C2 each; all outputs must stay identical.
# gpurun --timeout 1500 -- 'bash profiles/scripts/gpu_r6f.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6f; mkdir -p $O
export TMPDIR=/tmp
W=/tmp/w6f; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
TH=8
run() {   # tag, env...
  tag=$1; shift
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$tag.json "$@" timeout 200 $BIN -t $TH --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  echo "$tag: $(md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_$tag.txt) rounds $(python3 -c "import json;d=json.load(open('$O/stats_$tag.json'));print(d['add_query']['rounds'], 'kernel_ms', int(d['add_query']['kernel_ms']), 'pass %.1f' % (d['phases_s']['assembled']-d['phases_s']['trimmed_ready']))") elapsed $SECONDS"
  rm -f $W/o_${tag}_*
}
run warm
run head
run ahead24 T4_QUERY_AHEAD=24
run ahead48 T4_QUERY_AHEAD=48
run ahead72 T4_QUERY_AHEAD=72
run pend16 T4_MAX_PENDING=16
run light1 T4_LIGHT_AHEAD=1
run wide3k T4_WIDE_MIN_HITS=3072
run wide6k T4_WIDE_MIN_HITS=6144
run ext8 T4_AQ_EXTEND_DEFER=8
run ext32 T4_AQ_EXTEND_DEFER=32
run win128 T4_WINDOW=128
TH=12; run t12
TH=6; run t6
TH=8; run head2
echo "elapsed $SECONDS"
