#!/bin/bash
# round 6, session q: final HEAD -- the whole GPU suite, smoke, config C2 twice, the first 2 M pairs of C3 with the default (fragile
# checks off) on this box.
# gpurun --timeout 1800 -- 'bash profiles/scripts/gpu_r6q.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6q; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt; tail -4 $O/gpu_tests.txt | cut -c1-300
echo "elapsed $SECONDS"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/smoke.txt; tail -2 $O/smoke.txt | cut -c1-300
W=/tmp/w6q; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
for tag in c2_first c2_second; do
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$tag.json timeout 200 $BIN -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  echo "$tag: $(md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_$tag.txt) (17170ea8 47439b23 expected) elapsed $SECONDS"
done
rm -f $W/c2_* $W/o_c2*
tools/t4synth $W/ref.fa 2000000 200000 2 $W/c3 > /dev/null
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c3p2.json timeout 500 $BIN -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c3_1.fq -2 $W/c3_2.fq -o $W/o_c3 ) > $O/log_c3p2.txt 2>&1
echo "c3p2: $(md5sum $W/o_c3_raw.out $W/o_c3_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_c3p2.txt) (0c66030a 6f000dd4 expected) elapsed $SECONDS"
