#!/bin/bash
# round 6, session b: the query call without copy engines (aqPrologueKernel / aqEpilogueKernel, deltaKernel reading pinned memory) --
# config C2 with the section profile of the chain's host thread; query lanes again now that an invalidated entry needs a restricted
# re-query only (T4_LIVE_LANES 2 / 3); the GPU tests that force the overflow paths of the call.
# gpurun --timeout 1500 -- 'bash profiles/scripts/gpu_r6b.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6b; mkdir -p $O
export TMPDIR=/tmp
W=/tmp/w6b; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
ARGS="-t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq"
run() {   # tag, env...
  tag=$1; shift
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$tag.json "$@" timeout 200 $BIN $ARGS -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  echo "$tag: $(md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_$tag.txt) (17170ea8 47439b23 expected) elapsed $SECONDS"
}
run head
run head2
run lanes2 T4_LIVE_LANES=2
run lanes3 T4_LIVE_LANES=3
run lanes2b T4_LIVE_LANES=2 T4_LIVE_MIN_BATCH=1
timeout 600 python -m pytest tests/test_stage1_e2e.py -m gpu -x -q -k "bulk_live_set_paths_gpu or window_validity_rules_gpu or candidate_store_gpu or synthetic_matches" > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt | cut -c1-300
echo "elapsed $SECONDS"
