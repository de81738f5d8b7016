#!/bin/bash
# round 5, session f: thresholds with the light-round engine -- from how many emitted hits a read goes wide (the fresh heavy reads now
# start on the wide pipeline beside the query kernel), from how many overlaps its extensions go to extendKernel, contigs an entry may
# wait for, how far ahead whole queries reach, host threads.
# gpurun --timeout 1200 -- 'bash profiles/scripts/gpu_r5f.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5f; mkdir -p $O
W=/tmp/w5f; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
export TMPDIR=/tmp
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
run() { tag=$1; pre=$2; th=$3; shift 3
  ( time env T4_TIMING=1 "$@" T4_STATS_JSON=$O/stats_$tag.json timeout 300 trust4_amd/bin/trust4-hip -t $th --skipMateExtension -f $W/ref.fa -1 $W/${pre}_1.fq -2 $W/${pre}_2.fq -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-32 | tr '\n' ' ' >> $O/log_$tag.txt
  echo "== $tag: $(grep -h 'real' $O/log_$tag.txt | tr '\n' ' ') $(tail -1 $O/log_$tag.txt)"
  grep -h "assembler host seconds" $O/log_$tag.txt | cut -c1-420
  python3 -c "import json; d=json.load(open('$O/stats_$tag.json')); a=d['add_query']; p=d['phases_s']; c=d['chain']; print('rounds %d light %d whole %d restricted %d kernel_s %.1f hits %.1fG pass %.1f' % (c['rounds'], c['restricted_only_rounds'], c['whole_queries'], c['restricted_queries'], a['kernel_ms']/1e3, a['hits']/1e9, p['assembled']-p['trimmed_ready']))"
  rm -f $W/o_${tag}_*; }
run c2_combo c2 8 T4_WIDE_MIN_HITS=4096 T4_AQ_EXTEND_DEFER=16 T4_MAX_PENDING=8
run c2_w3k c2 8 T4_WIDE_MIN_HITS=3072 T4_AQ_EXTEND_DEFER=16 T4_MAX_PENDING=8
run c2_w2k c2 8 T4_WIDE_MIN_HITS=2048 T4_AQ_EXTEND_DEFER=16 T4_MAX_PENDING=8
run c2_w4k_only c2 8 T4_WIDE_MIN_HITS=4096
run c2_combo_t16 c2 16 T4_WIDE_MIN_HITS=4096 T4_AQ_EXTEND_DEFER=16 T4_MAX_PENDING=8
run c2_combo_t32 c2 32 T4_WIDE_MIN_HITS=4096 T4_AQ_EXTEND_DEFER=16 T4_MAX_PENDING=8
run c2_combo_a4 c2 8 T4_WIDE_MIN_HITS=4096 T4_AQ_EXTEND_DEFER=16 T4_MAX_PENDING=8 T4_AHEAD_MULT=4
run c2_combo_a2 c2 8 T4_WIDE_MIN_HITS=4096 T4_AQ_EXTEND_DEFER=16 T4_MAX_PENDING=8 T4_AHEAD_MULT=2
echo "elapsed $SECONDS  (C2: 17170ea8... 47439b23... expected)"
run b_combo b 8 T4_WIDE_MIN_HITS=4096 T4_AQ_EXTEND_DEFER=16 T4_MAX_PENDING=8
run b_w2k b 8 T4_WIDE_MIN_HITS=2048 T4_AQ_EXTEND_DEFER=16 T4_MAX_PENDING=8
echo "elapsed $SECONDS  (100 k pairs: 3d5fdf87... b4c66389... expected)"
