#!/bin/bash
# round 5, session e: posting marks (restricted re-queries off the contig), UpdateAllConsensus on touched contigs, light rounds as
# the default -- parity test, C2 with A/B switches, the phase-timing variant on the 100 k batch, ProcessRead on the device.
# gpurun --timeout 1500 -- 'bash profiles/scripts/gpu_r5e.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5e; mkdir -p $O
W=/tmp/w5e; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_stage1_e2e.py -m gpu -q -k "candidate_store" -s > $O/gpu_tests_cands.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_cands.txt; tail -3 $O/gpu_tests_cands.txt | cut -c1-400; grep "^('" $O/gpu_tests_cands.txt | cut -c1-300
echo "elapsed $SECONDS"
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
run() { tag=$1; pre=$2; shift 2
  ( time env T4_TIMING=1 "$@" T4_STATS_JSON=$O/stats_$tag.json timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/${pre}_1.fq -2 $W/${pre}_2.fq -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-32 | tr '\n' ' ' >> $O/log_$tag.txt
  echo "== $tag: $(grep -h 'real' $O/log_$tag.txt | tr '\n' ' ') $(tail -1 $O/log_$tag.txt)"
  grep -h "GPU query rounds\|assembler host seconds" $O/log_$tag.txt | cut -c1-420
  python3 -c "import json; d=json.load(open('$O/stats_$tag.json')); a=d['add_query']; p=d['phases_s']; print(d.get('chain'), 'kernel_s %.1f hits %.1fG pass %.1f input %.2f' % (a['kernel_ms']/1e3, a['hits']/1e9, p['assembled']-p['trimmed_ready'], p['input_processed_counted']))"
  rm -f $W/o_${tag}_*; }
run b_def b T4_X=1
( time env LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/trust4_amd/variants/phase T4_PHASE_DUMP=1 T4_TIMING=1 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/o_ph ) > $O/log_b_phase.txt 2>&1
grep -h "^phase\|AddRead queries with\|^   \|debug counters\|real" $O/log_b_phase.txt | cut -c1-200 | head -80
echo "elapsed $SECONDS"
run c2_def c2 T4_X=1
run c2_nomarks c2 T4_NO_MARKS=1
run c2_ed16 c2 T4_AQ_EXTEND_DEFER=16
run c2_ed32 c2 T4_AQ_EXTEND_DEFER=32
run c2_w4k c2 T4_WIDE_MIN_HITS=4096
run c2_mp8 c2 T4_MAX_PENDING=8
run c2_gpupr c2 T4_GPU_PROCESSREAD=1
echo "elapsed $SECONDS  (C2: 17170ea8... 47439b23... expected)"
