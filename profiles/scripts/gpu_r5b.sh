#!/bin/bash
# round 5, session b (VERDICT r4 "next" #1): the evidence HEAD lacks, on one MI355X --
#   (1) C2 through trust4-hip -t 8 with the phase seconds (the round's starting point),
#   (2) rocprofv3 --kernel-trace --stats of the benched 100 k-pair step and of C2 itself,
#   (3) the first 2 M pairs of C3 against the committed digest (bench.config_leg c3p2), counters at 0.5 / 1 / 2 M from the round log,
#   (4) the C5 recipe at 5 M pairs / 50 k cells (-t 32), md5 sums of the files (the reference's digest is made in the builder's container).
# gpurun --timeout 1800 -- 'bash profiles/scripts/gpu_r5b.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5b; mkdir -p $O
W=/tmp/w5b; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
export TMPDIR=/tmp
top() { python3 - "$1" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("%-70s calls %7s total %9.1f ms avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
}
# (1) C2
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c2.json timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/c2o ) > $O/log_c2.txt 2>&1
md5sum $W/c2o_raw.out $W/c2o_assembled_reads.fa | cut -c1-32 | tr '\n' ' ' >> $O/log_c2.txt
echo "== c2: $(grep -h 'real' $O/log_c2.txt | tr '\n' ' ') $(tail -1 $O/log_c2.txt) (17170ea8... 47439b23... expected)"
grep -h "assembler host seconds\|GPU query rounds\|wide query\|restricted" $O/log_c2.txt | cut -c1-600
echo "elapsed $SECONDS"
# (2) kernel traces
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b -o p -- $GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/pb ) > $O/prof_b.log 2>&1
f=$(find $O/prof_b -name "*kernel_stats.csv" | head -1); cp "$f" $O/r05_step_kernel_stats.csv; rm -rf $O/prof_b; echo "-- 100 k step"; top $O/r05_step_kernel_stats.csv
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o p -- $GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/pc2 ) > $O/prof_c2.log 2>&1
f=$(find $O/prof_c2 -name "*kernel_stats.csv" | head -1); cp "$f" $O/r05_c2_kernel_stats.csv; rm -rf $O/prof_c2; echo "-- C2"; top $O/r05_c2_kernel_stats.csv
echo "elapsed $SECONDS"
# (4) C5 recipe at 5 M pairs / 50 k cells
tools/t4synth $W/ref.fa 5000000 0 4 $W/c5 --cells 50000 > /dev/null
md5sum $W/c5_1.fq $W/c5_2.fq $W/c5_bc.fa $W/c5_umi.fa | cut -c1-32 | tr '\n' ' ' > $O/c5_inputs_md5.txt
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c5_5m.json timeout 400 trust4_amd/bin/trust4-hip -t 32 -f $W/ref.fa -1 $W/c5_1.fq -2 $W/c5_2.fq --barcode $W/c5_bc.fa --UMI $W/c5_umi.fa -o $W/c5o ) > $O/log_c5_5m.txt 2>&1
md5sum $W/c5o_raw.out $W/c5o_final.out $W/c5o_assembled_reads.fa | cut -c1-32 | tr '\n' ' ' >> $O/log_c5_5m.txt
echo "== c5 5M: $(grep -h 'real' $O/log_c5_5m.txt | tr '\n' ' ') $(tail -1 $O/log_c5_5m.txt)"; python3 -c "import json;print(json.load(open('$O/stats_c5_5m.json'))['phases_s'])"
rm -f $W/c5_* $W/c5o_*
echo "elapsed $SECONDS"
# (3) first 2 M pairs of C3 (the "Processed N reads" stamps of the log are the depth curve)
tools/t4synth $W/ref.fa 2000000 200000 2 $W/c3 > /dev/null
md5sum $W/c3_1.fq $W/c3_2.fq | cut -c1-32 | tr '\n' ' ' > $O/c3p2_inputs_md5.txt
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c3p2.json timeout 1100 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c3_1.fq -2 $W/c3_2.fq -o $W/c3o ) > $O/log_c3p2.txt 2>&1
md5sum $W/c3o_raw.out $W/c3o_final.out $W/c3o_assembled_reads.fa | cut -c1-32 | tr '\n' ' ' >> $O/log_c3p2.txt
python3 - $O <<'PY'
import json, sys
O = sys.argv[1]
g = json.load(open("tests/golden/c2_digests.json"))["c3p2"]
got = open(O + "/log_c3p2.txt").read().strip().split("\n")[-1].split()
want = [g["modes"]["skipMateExtension"]["md5"][x] for x in ("_raw.out", "_final.out", "_assembled_reads.fa")]
inp = open(O + "/c3p2_inputs_md5.txt").read().split()
print("== c3p2: inputs identical", inp == g["inputs_md5"], "outputs identical", got == want, got, want)
PY
grep -h "real\|Processed\|assembler host seconds\|GPU query rounds\|wide query\|Finish" $O/log_c3p2.txt | cut -c1-400
echo "elapsed $SECONDS"
