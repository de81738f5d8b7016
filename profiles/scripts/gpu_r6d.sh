#!/bin/bash
# round 6, session d: the wide pipeline behind the query kernel launched only when the kernel deferred a read (A/B: T4_WIDE_EAGER=1),
# sub-sections of the extension block of AddRead. Config C2, three runs; the wide-query GPU tests.
# gpurun --timeout 900 -- 'bash profiles/scripts/gpu_r6d.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6d; mkdir -p $O
export TMPDIR=/tmp
W=/tmp/w6d; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
ARGS="-t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq"
run() {   # tag, env...
  tag=$1; shift
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$tag.json "$@" timeout 200 $BIN $ARGS -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  echo "$tag: $(md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_$tag.txt) (17170ea8 47439b23 expected) elapsed $SECONDS"
}
run warm
run lazy
run eager T4_WIDE_EAGER=1
run lazy2
timeout 400 python -m pytest tests/test_wide_query.py tests/test_stage1_e2e.py -m gpu -x -q -k "wide or bulk_live_set_paths_gpu" > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt | cut -c1-300
echo "elapsed $SECONDS"
