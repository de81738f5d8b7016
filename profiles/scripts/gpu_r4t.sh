#!/bin/bash
# round 4, session t: HEAD after the cell groups and the host-phase work: barcode mode at 1 M pairs / 10 k cells (-t 32, defaults: 4 groups)
# against round 3's digests, the GPU e2e tests that cover what changed (cell groups, barcode mode, example, synthetic bulk, edge inputs, options)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4t; mkdir -p $O
W=/tmp/w4t; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 0 4 $W/c5 --cells 10000 > /dev/null
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c5.json timeout 100 trust4_amd/bin/trust4-hip -t 32 -f $W/ref.fa -1 $W/c5_1.fq -2 $W/c5_2.fq --barcode $W/c5_bc.fa --UMI $W/c5_umi.fa -o $W/c5o ) > $O/log_c5.txt 2>&1
md5sum $W/c5o_raw.out $W/c5o_assembled_reads.fa $W/c5o_final.out | cut -c1-32 | tr '\n' ' ' >> $O/log_c5.txt
grep -v "Read in and count" $O/log_c5.txt | cut -c1-250
echo "elapsed $SECONDS"
timeout 75 python -m pytest tests/test_stage1_e2e.py -m gpu -x -q -k "cell_groups_gpu or barcode_mode_matches or example_matches or synthetic_matches or edge_inputs or driver_options" > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -4 $O/pytest.txt
echo "elapsed $SECONDS"
