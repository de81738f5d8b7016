#!/bin/bash
# round 5, session a (prepared at the end of round 4, whose GPU minutes ran out before these could be taken): the state of HEAD on one
# MI355X -- the whole GPU suite (the early shard of --cellShard and two cell groups at -t 1 run on a GPU for the first time there),
# smoke, C2 through trust4-hip -t 8 against its digest with the phase seconds, barcode mode at 1 M pairs / 10 k cells (-t 32) twice
# (first run of a box / second), and the same with 8 cells' reads per round (T4_WINDOW) and 16 384 cells in flight.
# About 9 GPU-minutes. gpurun --timeout 900 -- 'bash profiles/scripts/gpu_r5a.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5a; mkdir -p $O
timeout 500 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
echo "elapsed $SECONDS"
W=/tmp/w5a; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 0 4 $W/c5 --cells 10000 > /dev/null
cells() { tag=$1; shift
  ( time env T4_TIMING=1 "$@" T4_STATS_JSON=$O/stats_c5_$tag.json timeout 60 trust4_amd/bin/trust4-hip -t 32 -f $W/ref.fa -1 $W/c5_1.fq -2 $W/c5_2.fq --barcode $W/c5_bc.fa --UMI $W/c5_umi.fa -o $W/c5o ) > $O/log_c5_$tag.txt 2>&1
  md5sum $W/c5o_raw.out $W/c5o_assembled_reads.fa | cut -c1-32 | tr '\n' ' ' >> $O/log_c5_$tag.txt
  echo "== c5 $tag: $(grep -h 'real\|^sys' $O/log_c5_$tag.txt | tr '\n' ' ') $(tail -1 $O/log_c5_$tag.txt) (57cc18cd... 89b90b07... expected)"; python3 -c "import json;print(json.load(open('$O/stats_c5_$tag.json'))['phases_s'])"; }
cells first T4_X=1
cells second T4_X=1
cells window8 T4_WINDOW=8
cells lanes16k T4_LANES=16384
echo "elapsed $SECONDS"
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c2.json timeout 200 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/c2o ) > $O/log_c2.txt 2>&1
md5sum $W/c2o_raw.out $W/c2o_assembled_reads.fa | cut -c1-32 | tr '\n' ' ' >> $O/log_c2.txt
echo "== c2: $(grep -h 'real\|^sys' $O/log_c2.txt | tr '\n' ' ') $(tail -1 $O/log_c2.txt) (17170ea8... 47439b23... expected)"; python3 -c "import json;print(json.load(open('$O/stats_c2.json'))['phases_s'])"
grep -h "assembler host seconds\|GPU query rounds" $O/log_c2.txt | cut -c1-400
echo "elapsed $SECONDS"
