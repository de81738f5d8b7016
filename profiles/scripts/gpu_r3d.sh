#!/bin/bash
# round 3, session d: GPU tests (seed replay over bit masks, traceback skip), 100 k pairs timing, per-phase profile
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
W=/tmp/w3d; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
( time env T4_TIMING=1 timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/m100 ) > $O/log_100k.txt 2>&1
md5sum $W/m100_raw.out >> $O/log_100k.txt
( time env T4_TIMING=1 T4_PHASE_DUMP=1 LD_LIBRARY_PATH=$PWD/trust4_amd/variants/phases timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/mph ) > $O/log_100k_phases.txt 2>&1
md5sum $W/mph_raw.out >> $O/log_100k_phases.txt
tail -3 $O/gpu_tests.txt; grep -h "real\|raw.out" $O/log_100k.txt; grep "phase \|debug counters\|real" $O/log_100k_phases.txt | tail -60
