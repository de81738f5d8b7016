#!/bin/bash
# round 5, session d: the candidate store with the exact replay of the group statistics, hulls from the restricted kernel, parallel
# copies of the dependency records, no extendKernel behind restricted-only rounds, hints for fresh heavy reads -- parity tests, then
# the light-round policy (T4_LIGHT_AHEAD = how near the head a whole query must be to ride with a head that waits for a restricted one).
# gpurun --timeout 1500 -- 'bash profiles/scripts/gpu_r5d.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5d; mkdir -p $O
W=/tmp/w5d; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_stage1_e2e.py -m gpu -q -k "candidate_store or stable_group_statistics" -s > $O/gpu_tests_cands.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_cands.txt; tail -4 $O/gpu_tests_cands.txt | cut -c1-400; grep "^('" $O/gpu_tests_cands.txt | cut -c1-300
echo "elapsed $SECONDS"
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
run() { tag=$1; pre=$2; shift 2
  ( time env T4_TIMING=1 "$@" T4_STATS_JSON=$O/stats_$tag.json timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/${pre}_1.fq -2 $W/${pre}_2.fq -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-32 | tr '\n' ' ' >> $O/log_$tag.txt
  echo "== $tag: $(grep -h 'real' $O/log_$tag.txt | tr '\n' ' ') $(tail -1 $O/log_$tag.txt)"
  grep -h "GPU query rounds\|candidate store\|query lanes\|assembler host seconds" $O/log_$tag.txt | cut -c1-420
  python3 -c "import json; d=json.load(open('$O/stats_$tag.json')); print(d.get('chain'), 'kernel_ms', d['add_query']['kernel_ms'])"
  rm -f $W/o_${tag}_*; }
run b_none b T4_X=1
run b_l0 b T4_LIGHT_AHEAD=0
run b_l1 b T4_LIGHT_AHEAD=1
run b_l2 b T4_LIGHT_AHEAD=2
run b_l4 b T4_LIGHT_AHEAD=4
echo "elapsed $SECONDS  (100 k pairs: 3d5fdf87... b4c66389... expected)"
run c2_l0 c2 T4_LIGHT_AHEAD=0
run c2_l1 c2 T4_LIGHT_AHEAD=1
run c2_l2 c2 T4_LIGHT_AHEAD=2
run c2_l2np c2 T4_LIGHT_AHEAD=2 T4_NO_PREDICT=1
echo "elapsed $SECONDS  (C2: 17170ea8... 47439b23... expected)"
( cd /tmp && T4_LIGHT_AHEAD=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o p -- $GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/pc2 ) > $O/prof_c2.log 2>&1
f=$(find $O/prof_c2 -name "*kernel_stats.csv" | head -1); cp "$f" $O/r05d_c2_l1_kernel_stats.csv; rm -rf $O/prof_c2
python3 - $O/r05d_c2_l1_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("%-70s calls %7s total %9.1f ms avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
echo "elapsed $SECONDS"
