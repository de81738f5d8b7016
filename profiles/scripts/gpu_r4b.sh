#!/bin/bash
# round 4, session b: where the wide query's time goes -- rocprofv3 kernel statistics of one 100 k-pair step (wide on / off), stats JSON (kernel ms)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p $O
W=/tmp/w4b; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
export TMPDIR=/tmp
for mode in on off; do
  if [ $mode = off ]; then export T4_WIDE_OFF=1; fi
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$mode.json trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/m_$mode ) > $O/log_$mode.txt 2>&1
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$mode -o p -- $GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/p_$mode ) > $O/prof_$mode.log 2>&1
  f=$(find $O/prof_$mode -name "*kernel_stats.csv" | head -1)
  echo "== wide $mode: $(grep real $O/log_$mode.txt)"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("%-70s calls %7s total %9.1f ms avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
  cp "$f" $O/kernel_stats_$mode.csv 2>/dev/null
  rm -rf $O/prof_$mode
done
python3 -c "
import json
for m in ('on','off'):
    s=json.load(open('$O/stats_%s.json'%m)); print(m, s['add_query'])"
