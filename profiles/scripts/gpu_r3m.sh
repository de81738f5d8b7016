#!/bin/bash
# round 3, session m: rough-annotation pass (2 M reads, resident) with the libraries of six commits on ONE box: where did 313 -> 365 ms come from?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3m; mkdir -p $O
for v in v_lean_field c_3cf3793; do
  if [ "$v" != head ]; then export T4_LIB=$PWD/trust4_amd/variants/$v/libt4hip.so; else unset T4_LIB; fi
  python tools/gpu_pass.py 2000000 4 2>&1 | tail -1 | sed "s/^/annotate pass [$v]: /" | tee -a $O/annotate_ab.txt
done
