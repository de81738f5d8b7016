#!/bin/bash
# round 5, session i: the last look at HEAD -- the whole GPU suite, smoke, C2, and barcode mode at 1 M pairs / 10 k cells (the k-mer
# count tables now start small and grow on the device).
# gpurun --timeout 1000 -- 'bash profiles/scripts/gpu_r5i.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5i; mkdir -p $O
W=/tmp/w5i; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt; tail -4 $O/gpu_tests.txt | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt | cut -c1-300
echo "elapsed $SECONDS"
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c2.json timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/c2o ) > $O/log_c2.txt 2>&1
md5sum $W/c2o_raw.out $W/c2o_assembled_reads.fa | cut -c1-32 | tr '\n' ' ' >> $O/log_c2.txt
echo "== c2: $(grep -h 'real' $O/log_c2.txt | tr '\n' ' ') $(tail -1 $O/log_c2.txt) (17170ea8... 47439b23... expected)"
grep -h "21-mers counted\|count statistics" $O/log_c2.txt | cut -c1-200
python3 -c "import json; d=json.load(open('$O/stats_c2.json')); a=d['add_query']; p=d['phases_s']; print(p, 'kernel_s %.1f' % (a['kernel_ms']/1e3))"
tools/t4synth $W/ref.fa 1000000 0 4 $W/c5 --cells 10000 > /dev/null
for tag in first second; do
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c5_$tag.json timeout 60 trust4_amd/bin/trust4-hip -t 32 -f $W/ref.fa -1 $W/c5_1.fq -2 $W/c5_2.fq --barcode $W/c5_bc.fa --UMI $W/c5_umi.fa -o $W/c5o ) > $O/log_c5_$tag.txt 2>&1
  md5sum $W/c5o_raw.out $W/c5o_assembled_reads.fa | cut -c1-32 | tr '\n' ' ' >> $O/log_c5_$tag.txt
  echo "== c5 1M $tag: $(grep -h 'real' $O/log_c5_$tag.txt | tr '\n' ' ') $(tail -1 $O/log_c5_$tag.txt) (57cc18cd... 89b90b07... expected)"; python3 -c "import json;print(json.load(open('$O/stats_c5_$tag.json'))['phases_s'])"
done
echo "elapsed $SECONDS"
