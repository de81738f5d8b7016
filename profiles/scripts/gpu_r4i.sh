#!/bin/bash
# round 4, session i: per-kernel time of the first 250 k pairs of C3 with the wide query from 8192 hits; the threshold below the LDS tier's capacity (4096, 2048); 100 k pairs shallow check
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r4i; mkdir -p $O
W=/tmp/w4i; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
export TMPDIR=/tmp
run() {
  local name=$1 pre=$2 lim=$3; shift; shift; shift
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$name.json "$@" timeout $lim trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 ${pre}_1.fq -2 ${pre}_2.fq -o $W/m_$name ) > $O/log_$name.txt 2>&1
  md5sum $W/m_${name}_raw.out $W/m_${name}_assembled_reads.fa >> $O/log_$name.txt
  echo "== $name: $(grep -h 'real' $O/log_$name.txt | tr '\n' ' ') $(grep -o 'GPU query rounds [0-9]* with [0-9]* reads' $O/log_$name.txt) $(grep -o 'wide query served [0-9]* window entries' $O/log_$name.txt)"
  grep -o '"kernel_ms": [0-9.]*' $O/stats_$name.json | tail -1
  tail -2 $O/log_$name.txt | cut -c1-34
}
tools/t4synth $W/ref.fa 250000 200000 2 $W/c3 > /dev/null
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- $GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c3_1.fq -2 $W/c3_2.fq -o $W/p ) > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c3p025_w8k.csv; rm -rf $O/prof
python3 - $O/kernel_stats_c3p025_w8k.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:10]:
    print("%-70s calls %7s total %9.1f ms avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
run c3_w4k $W/c3 600 T4_WIDE_MIN_HITS=4096
run c3_w2k $W/c3 600 T4_WIDE_MIN_HITS=2048
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
run b_w8k $W/b 300
run b_w4k $W/b 300 T4_WIDE_MIN_HITS=4096
run b_off $W/b 300 T4_WIDE_OFF=1
