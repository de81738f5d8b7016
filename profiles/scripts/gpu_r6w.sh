#!/bin/bash
# round 6, session w: T4_FRAGILE_CHECKS (entries of reads with lists beyond 10000 postings checked instead of falling to every edit) at a
# depth where such reads occur: the first 1 M pairs of C3. Default against the switch (md5 must agree), then the switch under
# T4_VERIFY_WINDOW (every served entry against a fresh whole query).
# gpurun --timeout 2400 -- 'bash profiles/scripts/gpu_r6w.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6w; mkdir -p $O
export TMPDIR=/tmp
W=/tmp/w6w; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
tools/t4synth $W/ref.fa 1000000 200000 2 $W/c3 > /dev/null
run() {   # tag, env...
  tag=$1; shift
  ( time env T4_TIMING=1 "$@" timeout 1500 $BIN -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c3_1.fq -2 $W/c3_2.fq -o $W/o_$tag ) > $O/log_$tag.txt 2>&1; echo "$tag rc $?"
  echo "$tag: $(md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_$tag.txt) elapsed $SECONDS"
  grep -h "thresholds checked\|tolerated index\|T4_VERIFY_WINDOW" $O/log_$tag.txt | cut -c1-330
}
run default
run fragile T4_FRAGILE_CHECKS=1
run fragile_verify T4_FRAGILE_CHECKS=1 T4_VERIFY_WINDOW=1
