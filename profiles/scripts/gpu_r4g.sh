#!/bin/bash
# round 4, session g: per-kernel time at depth (first 250 k pairs of C3 under rocprofv3 --kernel-trace --stats) and SQ counters of the AddRead query kernels / the rough-annotation tiers
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r4g; mkdir -p $O
W=/tmp/w4g; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
export TMPDIR=/tmp
tools/t4synth $W/ref.fa 250000 200000 2 $W/c3 > /dev/null
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c3p025.json trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c3_1.fq -2 $W/c3_2.fq -o $W/m ) > $O/log_c3p025.txt 2>&1; grep "real\|wide query\|restricted\|assembler host\|Finish assembly" $O/log_c3p025.txt | cut -c1-420
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- $GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c3_1.fq -2 $W/c3_2.fq -o $W/p ) > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c3p025.csv; rm -rf $O/prof
python3 - $O/kernel_stats_c3p025.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("%-70s calls %7s total %9.1f ms avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
# SQ counters (one pass each; --pmc with --kernel-trace only)
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  ( cd /tmp && rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/sq_$tag -o sq -- $GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/sq ) > $O/sq_$tag.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, collections, json, sys, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(sys.argv[1] + "/sq_*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
out = {k: dict(v) for k, v in agg.items() if "queryKernel" in k or "wide" in k or "extend" in k}
json.dump(out, open(sys.argv[1] + "/sq_counters.json", "w"), indent=1)
for k, v in out.items():
    wc, wait, act = v.get("SQ_WAVE_CYCLES", 0), v.get("SQ_WAIT_INST_ANY", 0), v.get("SQ_ACTIVE_INST_ANY", 0)
    tv, av = v.get("SQ_THREAD_CYCLES_VALU", 0), v.get("SQ_ACTIVE_INST_VALU", 0)
    print("%-60s waves %.3g wait/wavecycles %.2f valu lane use %.2f lds conflict/active %.3f" % (k, v.get("SQ_WAVES", 0), wait / wc if wc else 0, tv / (64 * av) if av else 0, v.get("SQ_LDS_BANK_CONFLICT", 0) / v.get("SQ_LDS_IDX_ACTIVE", 1) if v.get("SQ_LDS_IDX_ACTIVE") else 0))
PY
rm -rf $O/sq_SQ_*
