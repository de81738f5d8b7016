#!/bin/bash
# round 5, session m: the input dealt out by cells on the real engine -- four ranks sharing the box's GPU (file transport: ProcessRead
# and counts of a rank's own cells, t4_kmer_count_export / _merge on the device), one rank through RCCL (T4_SHARD_INPUT=2: the export
# and the all-gather of the table), the counter's export / merge against the oracle; then the C5 recipe at 1 M pairs through two ranks
# sharing the GPU against one process (what the exchange costs when nothing is gained: both ranks on one GPU and one host).
# gpurun --timeout 230 -- 'bash profiles/scripts/gpu_r5m.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5m; mkdir -p $O
export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_dist_gloo.py tests/test_zz_kmer_count_gpu.py tests/test_stage1_e2e.py -m gpu -q -x -k "engine_merge_four_ranks or two_rank_barcode or export_merge or rccl_gather_inside" > $O/gpu_tests_dealt.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_dealt.txt; tail -3 $O/gpu_tests_dealt.txt | cut -c1-300
echo "elapsed $SECONDS"
W=/tmp/w5m; mkdir -p $W/g; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 0 4 $W/c5 --cells 10000 > /dev/null
A="-f $W/ref.fa -1 $W/c5_1.fq -2 $W/c5_2.fq --barcode $W/c5_bc.fa --UMI $W/c5_umi.fa"
( time T4_TIMING=1 timeout 60 trust4_amd/bin/trust4-hip -t 32 $A -o $W/one ) > $O/log_one.txt 2>&1
for r in 0 1; do ( time T4_TIMING=1 timeout 60 trust4_amd/bin/trust4-hip -t 16 $A -o $W/two --cellShard $r/2 --gatherDir $W/g ) > $O/log_two_rank$r.txt 2>&1 & done; wait
md5sum $W/one_raw.out $W/two_raw.out $W/one_assembled_reads.fa $W/two_assembled_reads.fa | cut -c1-32 | tr '\n' ' '; echo
grep -h "real" $O/log_one.txt $O/log_two_rank0.txt $O/log_two_rank1.txt | tr '\n' ' '; echo
grep -h "counts of the ranks put together\|pairs of this rank's table\|their pairs alone" $O/log_two_rank*.txt | cut -c1-260
echo "elapsed $SECONDS"
