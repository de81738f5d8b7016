#!/bin/bash
# round 6, session v: the last committed HEAD -- the whole GPU suite, config C2 three times.
# gpurun --timeout 1200 -- 'bash profiles/scripts/gpu_r6v.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6v; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt; tail -4 $O/gpu_tests.txt | cut -c1-300
echo "elapsed $SECONDS"
W=/tmp/w6r; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
for tag in c2_first c2_second c2_third; do
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$tag.json timeout 200 $BIN -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  echo "$tag: $(md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_$tag.txt) (17170ea8 47439b23 expected) elapsed $SECONDS"
done
