#!/bin/bash
# round 3, final session of HEAD: the whole GPU suite, smoke, the default bench (steps + reference + C2 + side legs + PMC traffic), the
# kernel statistics of one step, C5 at 1 M pairs, C2 with default options through the bound reference main (trust4-dropin)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3q; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
tail -4 $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
T4_BENCH_PMC_DIR=$PWD/$O timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-1500 $O/bench.json
echo "elapsed after bench: $SECONDS s"
W=/tmp/w3q; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $W/prof -o step -- $GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/prof_run > /dev/null 2>&1 )
find $W/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/step_kernel_stats.csv; head -5 $O/step_kernel_stats.csv | cut -c1-200
echo "elapsed after kernel stats: $SECONDS s"; if [ $SECONDS -gt 960 ]; then echo "out of time: C5 and dropin legs skipped"; exit 0; fi
tools/t4synth $W/ref.fa 1000000 0 1 $W/c5 --cells 10000 > /dev/null
( time env T4_TIMING=1 T4_STATS_JSON=$O/stats_c5_t32.json timeout 600 trust4_amd/bin/trust4-hip -t 32 -f $W/ref.fa -1 $W/c5_1.fq -2 $W/c5_2.fq --barcode $W/c5_bc.fa --UMI $W/c5_umi.fa -o $W/c5o ) > $O/log_c5_t32.txt 2>&1
md5sum $W/c5o_raw.out $W/c5o_assembled_reads.fa >> $O/log_c5_t32.txt; grep "real" $O/log_c5_t32.txt; tail -2 $O/log_c5_t32.txt | cut -c1-34
echo "elapsed after C5: $SECONDS s"; if [ $SECONDS -gt 1000 ]; then echo "out of time: dropin leg skipped"; exit 0; fi
timeout 330 python bench.py --steps 1 --warmup 0 --c2 0 --cpu-baseline 0 --side-legs 0 --traffic 0 --config-leg c2:dropin > $O/bench_c2_dropin.json 2> $O/bench_c2_dropin.err; echo "dropin rc $?"; grep -o '"c2:dropin".\{0,400\}' $O/bench_c2_dropin.json
