#!/bin/bash
# round 5, session j: rocprofv3 --kernel-trace --stats of config C2 through HEAD (the per-kernel split behind the bench line's kernel_ms)
# gpurun --timeout 500 -- 'bash profiles/scripts/gpu_r5j.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5j; mkdir -p $O
W=/tmp/w5j; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
export TMPDIR=/tmp
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
( cd /tmp && T4_STATS_JSON=$O/stats_c2_traced.json timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o p -- $GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/pc2 ) > $O/prof_c2.log 2>&1
f=$(find $O/prof_c2 -name "*kernel_stats.csv" | head -1); cp "$f" $O/r05j_c2_kernel_stats_head.csv; rm -rf $O/prof_c2
python3 - $O/r05j_c2_kernel_stats_head.csv $O/stats_c2_traced.json <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("%-70s calls %7s total %9.1f ms avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
d = json.load(open(sys.argv[2]))
print("engine: rounds %d kernel_ms (HIP events, first launch .. last kernel of every round) %.1f" % (d["add_query"]["rounds"], d["add_query"]["kernel_ms"]))
PY
md5sum $W/pc2_raw.out $W/pc2_assembled_reads.fa | cut -c1-32 | tr '\n' ' '; echo "elapsed $SECONDS"
