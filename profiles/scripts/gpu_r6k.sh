#!/bin/bash
# round 6, session k: the emit mask -- an index edit reaches a window entry only at the k-mer positions GetHitsFromRead looks up and does not pass over; every booked edit is exact.
# statistics loop instead of a budget (A/B: T4_NO_EXACT_TOLERANCE=1). Config C2 interleaved; T4_VERIFY_WINDOW on two 30 k-pair inputs
# (600 clones: deep contigs; 15 000 clones: the group statistics live); the window / candidate-store GPU tests.
# gpurun --timeout 1500 -- 'bash profiles/scripts/gpu_r6k.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6k; mkdir -p $O
export TMPDIR=/tmp
W=/tmp/w6i; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
run() {   # tag, env...
  tag=$1; shift
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$tag.json "$@" timeout 200 $BIN -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  echo "$tag: $(md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_$tag.txt) rounds $(python3 -c "import json;d=json.load(open('$O/stats_$tag.json'));print(d['add_query']['rounds'], 'kernel_ms', int(d['add_query']['kernel_ms']), 'wait %.1f' % d['add_query']['host_wait_for_queries_s'], 'pass %.1f' % (d['phases_s']['assembled']-d['phases_s']['trimmed_ready']))") elapsed $SECONDS"
  rm -f $W/o_${tag}_*
}
run warm
run new1
run old1 T4_NO_EXACT_TOLERANCE=1
run new2
run old2 T4_NO_EXACT_TOLERANCE=1
tools/t4synth $W/ref.fa 30000 600 7 $W/v > /dev/null
( time env T4_VERIFY_WINDOW=1 T4_TIMING=1 timeout 400 $BIN -t 8 --skipMateExtension -f $W/ref.fa -1 $W/v_1.fq -2 $W/v_2.fq -o $W/o_v ) > $O/log_verify30k.txt 2>&1; echo "verify rc $?"; grep -h "T4_VERIFY_WINDOW\|real\|thresholds checked" $O/log_verify30k.txt | cut -c1-250
tools/t4synth $W/ref.fa 30000 15000 9 $W/v2 > /dev/null
( time env T4_VERIFY_WINDOW=1 T4_TIMING=1 timeout 400 $BIN -t 8 --skipMateExtension -f $W/ref.fa -1 $W/v2_1.fq -2 $W/v2_2.fq -o $W/o_v2 ) > $O/log_verify30k_manyclones.txt 2>&1; echo "verify rc $?"; grep -h "T4_VERIFY_WINDOW\|real\|thresholds checked" $O/log_verify30k_manyclones.txt | cut -c1-250
tools/t4synth $W/ref.fa 100000 50000 5 $W/v3 > /dev/null
( time env T4_VERIFY_WINDOW=1 T4_TIMING=1 timeout 600 $BIN -t 8 --skipMateExtension -f $W/ref.fa -1 $W/v3_1.fq -2 $W/v3_2.fq -o $W/o_v3 ) > $O/log_verify100k_manyclones.txt 2>&1; echo "verify rc $?"; grep -h "T4_VERIFY_WINDOW\|real\|thresholds checked" $O/log_verify100k_manyclones.txt | cut -c1-250
timeout 500 python -m pytest tests/test_stage1_e2e.py -m gpu -x -q -k "window or candidate_store_gpu" > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt | cut -c1-300
echo "elapsed $SECONDS"
