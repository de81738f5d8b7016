#!/bin/bash
# round 3, session j (pair DP with per-lane fixed problems): LDS-address-space DP loop + sort, scalar seed replay: GPU suite subset, 100 k timing, phase profile
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_lean_records.py tests/test_stage1_e2e.py -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
tail -3 $O/gpu_tests.txt
W=/tmp/w3j; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
for i in 1 2; do
( time env T4_TIMING=1 timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/m100 ) > $O/log_100k_$i.txt 2>&1
md5sum $W/m100_raw.out >> $O/log_100k_$i.txt
grep "real\|raw.out\|first launch" $O/log_100k_$i.txt
done
( time env T4_TIMING=1 T4_PHASE_DUMP=1 LD_LIBRARY_PATH=$PWD/trust4_amd/variants/phases timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/mph ) > $O/log_100k_phases.txt 2>&1
md5sum $W/mph_raw.out >> $O/log_100k_phases.txt
grep "phase .* lds\|debug counters\|real\|raw.out" $O/log_100k_phases.txt | tail -30
