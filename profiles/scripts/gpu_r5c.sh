#!/bin/bash
# round 5, session c: the candidate store on the GPU -- parity first (T4_VERIFY_WINDOW on many-clone inputs, wide query included),
# then what it buys: 100 k pairs and C2 with the store off (round 4's rule), on, and under the launch policies it makes possible
# (light rounds: a head that waits for a restricted re-query does not take the whole queries of entries far behind it along).
# gpurun --timeout 1500 -- 'bash profiles/scripts/gpu_r5c.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5c; mkdir -p $O
W=/tmp/w5c; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_stage1_e2e.py -m gpu -q -x -k "candidate_store or stable_group_statistics or window_validity_rules_gpu" -s > $O/gpu_tests_cands.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_cands.txt; tail -5 $O/gpu_tests_cands.txt | cut -c1-400
echo "elapsed $SECONDS"
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
run() { tag=$1; pre=$2; shift 2
  ( time env T4_TIMING=1 "$@" T4_STATS_JSON=$O/stats_$tag.json timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/${pre}_1.fq -2 $W/${pre}_2.fq -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-32 | tr '\n' ' ' >> $O/log_$tag.txt
  echo "== $tag: $(grep -h 'real' $O/log_$tag.txt | tr '\n' ' ') $(tail -1 $O/log_$tag.txt)"
  grep -h "GPU query rounds\|candidate store\|query lanes\|assembler host seconds" $O/log_$tag.txt | cut -c1-520
  rm -f $W/o_${tag}_*; }
run b_off b T4_CANDS_OFF=1
run b_on b T4_X=1
run b_l2 b T4_LIGHT_AHEAD=2
run b_l2h24 b T4_LIGHT_AHEAD=2 T4_HEAVY_BATCH=24 T4_AHEAD_MULT=5
run b_l0 b T4_LIGHT_AHEAD=0
run b_l6h32 b T4_LIGHT_AHEAD=6 T4_HEAVY_BATCH=32 T4_AHEAD_MULT=6
echo "elapsed $SECONDS  (100 k pairs: expected md5 of every run = the first's)"
run c2_off c2 T4_CANDS_OFF=1
run c2_on c2 T4_X=1
run c2_l2 c2 T4_LIGHT_AHEAD=2
run c2_l2h24 c2 T4_LIGHT_AHEAD=2 T4_HEAVY_BATCH=24 T4_AHEAD_MULT=5
run c2_l6h32 c2 T4_LIGHT_AHEAD=6 T4_HEAVY_BATCH=32 T4_AHEAD_MULT=6
echo "elapsed $SECONDS  (C2: 17170ea8... 47439b23... expected)"
