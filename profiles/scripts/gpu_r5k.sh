#!/bin/bash
# round 5, session k: a raised novelMinHitRequired served from the candidate store (run sizes ride with the candidate records) --
# T4_VERIFY_WINDOW on 30 k pairs of the C2 recipe's clone density (every served entry against a fresh whole query), the candidate-store
# parity test, then C2 timed once.
# gpurun --timeout 330 -- 'bash profiles/scripts/gpu_r5k.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5k; mkdir -p $O
W=/tmp/w5k; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
export TMPDIR=/tmp
tools/t4synth $W/ref.fa 30000 600 1 $W/v > /dev/null
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
run() { tag=$1; pre=$2; lim=$3; shift 3
  ( time env T4_TIMING=1 "$@" T4_STATS_JSON=$O/stats_$tag.json timeout $lim trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/${pre}_1.fq -2 $W/${pre}_2.fq -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-32 | tr '\n' ' ' >> $O/log_$tag.txt
  echo "== $tag: $(grep -h 'real' $O/log_$tag.txt | tr '\n' ' ') $(tail -1 $O/log_$tag.txt)"
  grep -h "GPU query rounds\|candidate store\|T4_VERIFY_WINDOW" $O/log_$tag.txt | cut -c1-900
  rm -f $W/o_${tag}_*; }
run v_plain v 60 T4_X=1
run v_verify v 100 T4_VERIFY_WINDOW=1
echo "elapsed $SECONDS"
run c2 c2 120 T4_X=1
echo "elapsed $SECONDS  (C2: 17170ea8... 47439b23... expected)"
timeout 100 python -m pytest tests/test_stage1_e2e.py -m gpu -q -k "candidate_store" > $O/gpu_tests_cands.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_cands.txt; tail -3 $O/gpu_tests_cands.txt | cut -c1-300
echo "elapsed $SECONDS"
