#!/bin/bash
# round 6, session u: the C5 recipe at 20 M pairs / 50 k cells (40 % of config C5's pairs over all of its cells) on ONE MI355X, barcode mode;
# the md5 sums are held against the reference's digest (tools/c2_digests.py --config c5m20, 2.7 h of the reference at -t 8 in the builder's container).
# gpurun --timeout 1500 -- 'bash profiles/scripts/gpu_r6u.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6u; mkdir -p $O
export TMPDIR=/tmp
W=/tmp/w6u; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
( time tools/t4synth $W/ref.fa 20000000 0 4 $W/c5 --cells 50000 > /dev/null ) 2>&1 | grep real
md5sum $W/c5_1.fq $W/c5_2.fq $W/c5_bc.fa $W/c5_umi.fa | cut -c1-32
echo "elapsed $SECONDS"
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c5m20.json timeout 900 $BIN -t 32 -f $W/ref.fa -1 $W/c5_1.fq -2 $W/c5_2.fq --barcode $W/c5_bc.fa --UMI $W/c5_umi.fa -o $W/o ) > $O/log_c5m20.txt 2>&1; echo "rc $?"
echo "c5m20: $(md5sum $W/o_raw.out $W/o_assembled_reads.fa $W/o_final.out | cut -c1-32 | tr '\n' ' ') $(grep -h real $O/log_c5m20.txt) elapsed $SECONDS"
ls -la $W/o_raw.out $W/o_assembled_reads.fa | awk '{print $5, $9}'
grep -h "Finish assembly\|timing: input" $O/log_c5m20.txt | cut -c1-300
