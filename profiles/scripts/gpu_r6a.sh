#!/bin/bash
# round 6, session a: round 5's HEAD on this round's box -- config C2 once (timing line), once with the round log (what a restricted-only
# round costs), and C2's HBM traffic measured ON C2 ITSELF: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over the whole run, per kernel.
# gpurun --timeout 3000 -- 'bash profiles/scripts/gpu_r6a.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6a; mkdir -p $O
export TMPDIR=/tmp
W=/tmp/w6a; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
ARGS="-t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq"
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c2.json timeout 150 $BIN $ARGS -o $W/o ) > $O/log_c2.txt 2>&1
md5sum $W/o_raw.out $W/o_assembled_reads.fa | cut -c1-32 | tr '\n' ' '; grep -h real $O/log_c2.txt; echo "(C2: 17170ea8... 47439b23... expected)"
echo "elapsed $SECONDS"
( time env T4_TIMING=1 T4_ROUND_LOG=$W/rounds.txt timeout 200 $BIN $ARGS -o $W/o2 ) > $O/log_c2_roundlog.txt 2>&1
gzip -c $W/rounds.txt > $O/rounds_c2.txt.gz; ls -la $O/rounds_c2.txt.gz
echo "elapsed $SECONDS"
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 1300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $W/pmc_$C -o pmc -- $BIN $ARGS -o $W/p$C ) > $O/pmc_$C.log 2>&1
  echo "pmc $C rc $? elapsed $SECONDS"
  f=$(find $W/pmc_$C -name "*counter_collection.csv" | head -1)
  python3 - "$f" $C $O/r06_c2_pmc_$C.txt <<'PY'
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
print(open(sys.argv[3]).read()[:1500])
PY
  md5sum $W/p${C}_raw.out | cut -c1-32
  rm -rf $W/pmc_$C
done
echo "elapsed $SECONDS"
