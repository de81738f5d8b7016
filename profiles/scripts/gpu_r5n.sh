#!/bin/bash
# round 5, session n: HEAD at the end of the round -- test_candidate_store_gpu with its third input (thresholds raised at depth, served from
# the store, every entry verified) and config C2 once through the binary as committed (md5 against the reference's digest).
# gpurun --timeout 200 -- 'bash profiles/scripts/gpu_r5n.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5n; mkdir -p $O
export TMPDIR=/tmp
W=/tmp/w5n; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c2.json timeout 100 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/o ) > $O/log_c2.txt 2>&1
md5sum $W/o_raw.out $W/o_assembled_reads.fa | cut -c1-32 | tr '\n' ' '; grep -h real $O/log_c2.txt; echo "(C2: 17170ea8... 47439b23... expected)"
echo "elapsed $SECONDS"
timeout 110 python -m pytest tests/test_stage1_e2e.py -m gpu -q -k "candidate_store" > $O/gpu_tests_cands.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_cands.txt; tail -3 $O/gpu_tests_cands.txt | cut -c1-300
echo "elapsed $SECONDS"
