#!/bin/bash
# round 3, session e: the tandem-repeat skip-rule test with the bit-mask replay and with the one-lane replay (variant), then the GPU suite, the phase profile
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3e; mkdir -p $O
T4_LIB=$PWD/trust4_amd/variants/seedserial/libt4hip.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k novel_min > $O/test_seedserial.txt 2>&1; echo "rc $?" >> $O/test_seedserial.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k novel_min > $O/test_new.txt 2>&1; echo "rc $?" >> $O/test_new.txt
tail -3 $O/test_seedserial.txt; tail -3 $O/test_new.txt
W=/tmp/w3e; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
( time env T4_TIMING=1 T4_PHASE_DUMP=1 LD_LIBRARY_PATH=$PWD/trust4_amd/variants/phases timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/mph ) > $O/log_100k_phases.txt 2>&1
md5sum $W/mph_raw.out >> $O/log_100k_phases.txt
grep "phase \|debug counters\|real\|raw.out" $O/log_100k_phases.txt | tail -70
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
tail -5 $O/gpu_tests.txt
