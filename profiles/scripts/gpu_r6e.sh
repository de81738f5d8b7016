#!/bin/bash
# round 6, session e: UpdateIndexFromRead through hints that carry the k-mer code (no look at the key map), host seconds inside the
# query call by section. Config C2 twice; the first 2 M pairs of C3 (200 k clones) against the reference's digest.
# gpurun --timeout 1200 -- 'bash profiles/scripts/gpu_r6e.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6e; mkdir -p $O
export TMPDIR=/tmp
W=/tmp/w6e; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
ARGS="-t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq"
run() {   # tag, env...
  tag=$1; shift
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$tag.json "$@" timeout 200 $BIN $ARGS -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  echo "$tag: $(md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_$tag.txt) (17170ea8 47439b23 expected) elapsed $SECONDS"
}
run head
run head2
tools/t4synth $W/ref.fa 2000000 200000 2 $W/c3 > /dev/null
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c3p2.json timeout 500 $BIN -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c3_1.fq -2 $W/c3_2.fq -o $W/o_c3 ) > $O/log_c3p2.txt 2>&1
echo "c3p2: $(md5sum $W/o_c3_raw.out $W/o_c3_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_c3p2.txt) (0c66030a 6f000dd4 expected) elapsed $SECONDS"
