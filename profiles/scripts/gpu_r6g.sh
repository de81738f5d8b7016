#!/bin/bash
# round 6, session g: window k-mer map in four shards filled side by side; the sample behind a wide read's partition plan (T4_WIDE_SAMPLE:
# hits per planned partition), the wide query from 3 072 emitted hits. Interleaved runs of config C2 (the boxes drift by seconds).
# gpurun --timeout 1200 -- 'bash profiles/scripts/gpu_r6g.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6g; mkdir -p $O
export TMPDIR=/tmp
W=/tmp/w6g; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
run() {   # tag, env...
  tag=$1; shift
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$tag.json "$@" timeout 200 $BIN -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  echo "$tag: $(md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_$tag.txt) rounds $(python3 -c "import json;d=json.load(open('$O/stats_$tag.json'));print(d['add_query']['rounds'], 'kernel_ms', int(d['add_query']['kernel_ms']), 'wait %.1f' % d['add_query']['host_wait_for_queries_s'], 'pass %.1f' % (d['phases_s']['assembled']-d['phases_s']['trimmed_ready']))") elapsed $SECONDS"
  rm -f $W/o_${tag}_*
}
run warm
run a1
run s256_1 T4_WIDE_SAMPLE=256
run w3k_1 T4_WIDE_MIN_HITS=3072
run both_1 T4_WIDE_MIN_HITS=3072 T4_WIDE_SAMPLE=256
run a2
run s256_2 T4_WIDE_SAMPLE=256
run w3k_2 T4_WIDE_MIN_HITS=3072
run both_2 T4_WIDE_MIN_HITS=3072 T4_WIDE_SAMPLE=256
run s128 T4_WIDE_MIN_HITS=3072 T4_WIDE_SAMPLE=128
run w2k T4_WIDE_MIN_HITS=2048 T4_WIDE_SAMPLE=256
echo "elapsed $SECONDS"
