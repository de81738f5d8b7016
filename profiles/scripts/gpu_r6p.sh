#!/bin/bash
# round 6, session p: window entries of reads with lists beyond 10000 postings (removeOnlyRepeats) no longer fall to every index edit:
# booked exactly and checked -- threshold, removeOnlyRepeats per strand, the head of the hit array (exactStats). T4_VERIFY_WINDOW on the
# first 500 k pairs of C3 (200 k clones: lists beyond 10000 postings appear there); the first 2 M pairs timed (A/B: T4_NO_EXACT_TOLERANCE);
# config C2; the wide-query GPU tests.
# gpurun --timeout 2400 -- 'bash profiles/scripts/gpu_r6p.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6p; mkdir -p $O
export TMPDIR=/tmp
W=/tmp/w6p; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
timeout 300 python -m pytest tests/test_wide_query.py -m gpu -x -q > $O/gpu_tests_wide.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_wide.txt; tail -2 $O/gpu_tests_wide.txt | cut -c1-200
tools/t4synth $W/ref.fa 500000 200000 2 $W/c3s > /dev/null
( time env T4_TIMING=1 timeout 300 $BIN -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c3s_1.fq -2 $W/c3s_2.fq -o $W/o_c3s ) > $O/log_c3p05.txt 2>&1
echo "c3p05: $(md5sum $W/o_c3s_raw.out $W/o_c3s_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_c3p05.txt) elapsed $SECONDS"; grep -h "thresholds checked\|tolerated index" $O/log_c3p05.txt | cut -c1-330
( time env T4_VERIFY_WINDOW=1 T4_TIMING=1 timeout 1500 $BIN -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c3s_1.fq -2 $W/c3s_2.fq -o $W/o_c3v ) > $O/log_c3p05_verify.txt 2>&1; echo "verify rc $? elapsed $SECONDS"
grep -h "T4_VERIFY_WINDOW\|real\|thresholds checked" $O/log_c3p05_verify.txt | cut -c1-330; md5sum $W/o_c3v_raw.out | cut -c1-8
rm -f $W/c3s_* $W/o_c3*
tools/t4synth $W/ref.fa 2000000 200000 2 $W/c3 > /dev/null
for tag in c3p2_new c3p2_old; do
  extra=""; [ $tag = c3p2_old ] && extra="T4_NO_EXACT_TOLERANCE=1"
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$tag.json $extra timeout 500 $BIN -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c3_1.fq -2 $W/c3_2.fq -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  echo "$tag: $(md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_$tag.txt) (0c66030a 6f000dd4 expected) elapsed $SECONDS"
  grep -h "thresholds checked\|tolerated index\|rounds whose head" $O/log_$tag.txt | cut -c1-400
  rm -f $W/o_${tag}_*
done
rm -f $W/c3_*
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c2.json timeout 200 $BIN -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/o_c2 ) > $O/log_c2.txt 2>&1
echo "c2: $(md5sum $W/o_c2_raw.out $W/o_c2_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_c2.txt) (17170ea8 47439b23 expected) elapsed $SECONDS"
