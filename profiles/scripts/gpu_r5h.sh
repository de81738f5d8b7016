#!/bin/bash
# round 5, session h: HEAD after the commit-path changes (candidate records in place, predicate flips without masks, no event per
# delta) -- the bulk / window GPU tests, C2 with 8 and with 4 records per extension wavefront (variant library), and C3's prefixes
# with the round-5 engine (500 k pairs, 2 M pairs against the committed digests).
# gpurun --timeout 1500 -- 'bash profiles/scripts/gpu_r5h.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5h; mkdir -p $O
W=/tmp/w5h; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_stage1_e2e.py tests/test_assembler_emu.py -m gpu -q > $O/gpu_tests_bulk.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_bulk.txt; tail -3 $O/gpu_tests_bulk.txt | cut -c1-300
echo "elapsed $SECONDS"
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
run() { tag=$1; pre=$2; shift 2
  ( time env T4_TIMING=1 "$@" T4_STATS_JSON=$O/stats_$tag.json timeout 900 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/${pre}_1.fq -2 $W/${pre}_2.fq -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  md5sum $W/o_${tag}_raw.out $W/o_${tag}_final.out $W/o_${tag}_assembled_reads.fa | cut -c1-32 | tr '\n' ' ' >> $O/log_$tag.txt
  echo "== $tag: $(grep -h 'real' $O/log_$tag.txt | tr '\n' ' ') $(tail -1 $O/log_$tag.txt)"
  grep -h "assembler host seconds" $O/log_$tag.txt | cut -c1-420
  python3 -c "import json; d=json.load(open('$O/stats_$tag.json')); a=d['add_query']; p=d['phases_s']; c=d['chain']; print('rounds %d light %d whole %d restricted %d kernel_s %.1f hits %.1fG pass %.1f deprec %.2fG' % (c['rounds'], c['restricted_only_rounds'], c['whole_queries'], c['restricted_queries'], a['kernel_ms']/1e3, a['hits']/1e9, p['assembled']-p['trimmed_ready'], a['wide']['dependency_records']/1e9))"
  rm -f $W/o_${tag}_*; }
run c2_head c2 T4_X=1
run c2_ext4 c2 LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/trust4_amd/variants/ext4
run c2_head2 c2 T4_X=1
echo "elapsed $SECONDS  (C2: 17170ea8... 17170ea8... 47439b23... expected)"
tools/t4synth $W/ref.fa 500000 200000 2 $W/c3a > /dev/null
run c3p05 c3a T4_X=1
echo "(c3p05: 2daa33ee... 2daa33ee... c4e2c8eb... expected)"
tools/t4synth $W/ref.fa 2000000 200000 2 $W/c3b > /dev/null
run c3p2 c3b T4_X=1
grep -h "Processed [1234]000000 reads\|Finish rough\|Assembled" $O/log_c3p2.txt | cut -c1-120
echo "(c3p2: 0c66030a... 0c66030a... 6f000dd4... expected)  elapsed $SECONDS"
