#!/bin/bash
# round 6, session t: -t beyond 8 with the chain's helpers capped at eight (the input phases take all of -t). Config C2, interleaved.
# gpurun --timeout 900 -- 'bash profiles/scripts/gpu_r6t.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6t; mkdir -p $O
export TMPDIR=/tmp
W=/tmp/w6t; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
BIN=$GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip
run() {   # tag, threads, env...
  tag=$1; th=$2; shift; shift
  ( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_$tag.json "$@" timeout 200 $BIN -t $th --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/o_$tag ) > $O/log_$tag.txt 2>&1
  echo "$tag: $(md5sum $W/o_${tag}_raw.out $W/o_${tag}_assembled_reads.fa | cut -c1-8 | tr '\n' ' ') $(grep -h real $O/log_$tag.txt) $(python3 -c "import json;d=json.load(open('$O/stats_$tag.json'));p=d['phases_s'];print('before the pass %.2f pass %.1f' % (p['trimmed_ready'], p['assembled']-p['trimmed_ready']))") elapsed $SECONDS"
  rm -f $W/o_${tag}_*
}
run warm 8
run t8a 8
run t16a 16
run t32a 32
run t8b 8
run t16b 16
run t32b 32
run t32_chain4 32 T4_CHAIN_THREADS=4
run t32_chain12 32 T4_CHAIN_THREADS=12
