#!/bin/bash
# round 6, session m: HEAD -- the whole GPU suite and smoke, config C2 twice (timing line, stats), rocprofv3 --kernel-trace --stats of C2,
# and C2's HBM-side traffic measured on C2 itself (FETCH_SIZE / WRITE_SIZE passes, per kernel).
# gpurun --timeout 2400 -- 'bash profiles/scripts/gpu_r6m.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6m; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt; tail -4 $O/gpu_tests.txt | cut -c1-300
echo "elapsed $SECONDS"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/smoke.txt; tail -2 $O/smoke.txt | cut -c1-400
echo "elapsed $SECONDS"
W=/tmp/w6m; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
ARGS="-t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq"
for tag in c2_first c2_second; do
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$tag.json timeout 200 $BIN $ARGS -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  echo "$tag: $(md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_$tag.txt) (17170ea8 47439b23 expected) elapsed $SECONDS"
done
( cd /tmp && T4_STATS_JSON=$O/stats_c2_traced.json timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $W/prof_c2 -o p -- $BIN $ARGS -o $W/pc2 ) > $O/prof_c2.log 2>&1
f=$(find $W/prof_c2 -name "*kernel_stats.csv" | head -1); cp "$f" $O/r06m_c2_kernel_stats.csv; rm -rf $W/prof_c2
python3 - $O/r06m_c2_kernel_stats.csv $O/stats_c2_traced.json <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print("%-70s calls %7s total %9.1f ms avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
d = json.load(open(sys.argv[2]))
print("engine: rounds %d kernel_ms (HIP events, first launch .. last kernel of every round) %.1f" % (d["add_query"]["rounds"], d["add_query"]["kernel_ms"]))
PY
echo "elapsed $SECONDS"
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $W/pmc_$C -o pmc -- $BIN $ARGS -o $W/p$C ) > $O/pmc_$C.log 2>&1
  echo "pmc $C rc $? elapsed $SECONDS"
  f=$(find $W/pmc_$C -name "*counter_collection.csv" | head -1)
  python3 - "$f" $C $O/r06m_c2_pmc_$C.txt <<'PY'
import csv, re, sys
acc = {}
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        if row.get("Counter_Name") != sys.argv[2]:
            continue
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).strip()
        a = acc.setdefault(name, [0, 0.0])
        a[0] += 1; a[1] += float(row["Counter_Value"])
with open(sys.argv[3], "w") as g:
    g.write("# rocprofv3 --pmc %s --kernel-trace over ONE WHOLE RUN of config C2 (1 M pairs) through trust4-hip -t 8 --skipMateExtension; counter units: KB; per kernel: launches, sum\n" % sys.argv[2])
    for name, (cnt, val) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        g.write("%-110s %8d %16.0f\n" % (name, cnt, val))
PY
  md5sum $W/p${C}_raw.out | cut -c1-8
  rm -rf $W/pmc_$C
done
echo "elapsed $SECONDS"
