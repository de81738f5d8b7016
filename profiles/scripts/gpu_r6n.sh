#!/bin/bash
# round 6, session n: how far behind the head should RESTRICTED re-queries reach? (the reads of a clone extend the same contig end one
# after the other: an entry far behind the head is touched again before it is served). T4_RESTRICT_AHEAD fixed / relative. Config C2.
# gpurun --timeout 1200 -- 'bash profiles/scripts/gpu_r6n.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6n; mkdir -p $O
export TMPDIR=/tmp
W=/tmp/w6n; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
run() {   # tag, env...
  tag=$1; shift
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$tag.json "$@" timeout 200 $BIN -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  echo "$tag: $(md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_$tag.txt) rounds $(python3 -c "import json;d=json.load(open('$O/stats_$tag.json'));print(d['add_query']['rounds'], 'light', d['chain']['restricted_only_rounds'], 'kernel_ms', int(d['add_query']['kernel_ms']), 'wait %.1f' % d['add_query']['host_wait_for_queries_s'], 'whole', d['chain']['whole_queries'], 'restricted', d['chain']['restricted_queries'], 'pass %.1f' % (d['phases_s']['assembled']-d['phases_s']['trimmed_ready']))") elapsed $SECONDS"
  rm -f $W/o_${tag}_*
}
run warm
run head1
run r4 T4_RESTRICT_AHEAD=4
run r8 T4_RESTRICT_AHEAD=8
run r12 T4_RESTRICT_AHEAD=12
run r16 T4_RESTRICT_AHEAD=16
run head2
run rel2 T4_RESTRICT_AHEAD=-2
run rel4 T4_RESTRICT_AHEAD=-4
run r8b T4_RESTRICT_AHEAD=8
run r12b T4_RESTRICT_AHEAD=12
echo "elapsed $SECONDS"
