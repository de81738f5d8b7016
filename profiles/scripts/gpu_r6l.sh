#!/bin/bash
# round 6, session l: with the tolerance kills gone, the heads that wait for a WHOLE query are the reads nobody has queried yet: how far
# ahead of the head should a whole-query round reach? (T4_QUERY_AHEAD, fixed; default 3 x reads served per round + 12 ~ 33). Config C2.
# gpurun --timeout 1200 -- 'bash profiles/scripts/gpu_r6l.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6l; mkdir -p $O
export TMPDIR=/tmp
W=/tmp/w6l; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
run() {   # tag, env...
  tag=$1; shift
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$tag.json "$@" timeout 200 $BIN -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  echo "$tag: $(md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_$tag.txt) rounds $(python3 -c "import json;d=json.load(open('$O/stats_$tag.json'));print(d['add_query']['rounds'], 'light', d['chain']['restricted_only_rounds'], 'kernel_ms', int(d['add_query']['kernel_ms']), 'wait %.1f' % d['add_query']['host_wait_for_queries_s'], 'whole', d['chain']['whole_queries'], 'pass %.1f' % (d['phases_s']['assembled']-d['phases_s']['trimmed_ready']))") elapsed $SECONDS"
  rm -f $W/o_${tag}_*
}
run warm
run head1
run a48 T4_QUERY_AHEAD=48
run a64 T4_QUERY_AHEAD=64
run a96 T4_QUERY_AHEAD=96
run a128 T4_QUERY_AHEAD=128
run head2
run a64b T4_QUERY_AHEAD=64
run a96b T4_QUERY_AHEAD=96
run a160 T4_QUERY_AHEAD=160 T4_WINDOW=256
run pend16 T4_QUERY_AHEAD=96 T4_MAX_PENDING=16
echo "elapsed $SECONDS"
