#!/bin/bash
# round 6, session o: HEAD with wideSeedKernel's argument struct in LDS (224 -> 32 bytes of scratch per lane). Config C2 twice and its
# two PMC passes again; the first 2 M pairs of C3 (depth curve); barcode mode: the C5 recipe at 1 M pairs / 10 k cells twice and at 5 M
# pairs / 50 k cells, all against the reference's digests; `python bench.py --steps 2 --warmup 1` (the whole flow of the driver's line).
# gpurun --timeout 3000 -- 'bash profiles/scripts/gpu_r6o.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6o; mkdir -p $O
export TMPDIR=/tmp
W=/tmp/w6o; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
ARGS="-t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq"
for tag in c2_first c2_second; do
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$tag.json timeout 200 $BIN $ARGS -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  echo "$tag: $(md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_$tag.txt) (17170ea8 47439b23 expected) elapsed $SECONDS"
done
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $W/pmc_$C -o pmc -- $BIN $ARGS -o $W/p$C ) > $O/pmc_$C.log 2>&1
  echo "pmc $C rc $? elapsed $SECONDS"
  f=$(find $W/pmc_$C -name "*counter_collection.csv" | head -1)
  python3 - "$f" $C $O/r06o_c2_pmc_$C.txt <<'PY'
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
  rm -rf $W/pmc_$C
done
tools/t4synth $W/ref.fa 2000000 200000 2 $W/c3 > /dev/null
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c3p2.json timeout 500 $BIN -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c3_1.fq -2 $W/c3_2.fq -o $W/o_c3 ) > $O/log_c3p2.txt 2>&1
echo "c3p2: $(md5sum $W/o_c3_raw.out $W/o_c3_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_c3p2.txt) (0c66030a 6f000dd4 expected) elapsed $SECONDS"
rm -f $W/c3_* $W/o_c3*
python3 - <<'PY'
import json; d = json.load(open("tests/golden/c2_digests.json"))
for k in ("c5_1m", "c5m5"):
    print(k, "expected", {s: v[:8] for s, v in d[k]["md5"].items()} if "md5" in d[k] else {s: v[:8] for s, v in d[k]["modes"]["barcode"]["md5"].items()})
PY
tools/t4synth $W/ref.fa 1000000 0 4 $W/c5 --cells 10000 > /dev/null
for tag in c5_1m_first c5_1m_second; do
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$tag.json timeout 200 $BIN -t 32 -f $W/ref.fa -1 $W/c5_1.fq -2 $W/c5_2.fq --barcode $W/c5_bc.fa --UMI $W/c5_umi.fa -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  echo "$tag: $(md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa $W/o_${tag}_final.out | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_$tag.txt) elapsed $SECONDS"
done
rm -f $W/c5_* $W/o_c5*
tools/t4synth $W/ref.fa 5000000 0 4 $W/c55 --cells 50000 > /dev/null
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c5m5.json timeout 400 $BIN -t 32 -f $W/ref.fa -1 $W/c55_1.fq -2 $W/c55_2.fq --barcode $W/c55_bc.fa --UMI $W/c55_umi.fa -o $W/o_c5m5 ) > $O/log_c5m5.txt 2>&1
echo "c5m5: $(md5sum $W/o_c5m5_raw.out $W/o_c5m5_assembled_reads.fa $W/o_c5m5_final.out | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_c5m5.txt) elapsed $SECONDS"
rm -rf $W
( time timeout 1200 python bench.py --steps 2 --warmup 1 ) > $O/bench_steps2.json 2> $O/bench_steps2.err; echo "bench rc $?"; tail -c 600 $O/bench_steps2.json; grep real $O/bench_steps2.err
echo "elapsed $SECONDS"
