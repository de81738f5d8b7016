#!/bin/bash
# round 3, session n: per-kernel times of the rough-annotation pass, HEAD (seedref build) against the round-2 library, same box
cd "$GRAFT_REPO_ROOT" || exit 1
O=$PWD/gpurun_out/r3n; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in seedref r2final; do
  T4_LIB=$R/trust4_amd/variants/$v/libt4hip.so rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o p -- python $R/tools/gpu_pass.py 2000000 3 > $O/pass_$v.txt 2>&1
  F=$(find $O/prof_$v -name "*kernel_stats*" | head -1); cp $F $O/kernel_stats_$v.csv; rm -rf $O/prof_$v
  echo "== $v"; tail -1 $O/pass_$v.txt; head -9 $O/kernel_stats_$v.csv | cut -c1-160
done
