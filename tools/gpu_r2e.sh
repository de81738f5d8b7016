mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
export T4_TIMING=1
D=/tmp/r2e; mkdir -p $D
zcat data/hg38_bcrtcr.fa.gz > $D/ref.fa
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_stage1_e2e.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r2e_tests.txt
cat gpurun_out/r2e_tests.txt
for N in 20000 100000; do
  tools/t4synth data/hg38_bcrtcr.fa.gz $N $((N/50)) 1 $D/s$N > /dev/null
  ( time T4_STATS_JSON=$R/gpurun_out/r2e_stats_$N.json trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s${N}_1.fq -2 $D/s${N}_2.fq -o $D/mine$N ) 2>&1 | tail -9 > gpurun_out/r2e_$N.txt
  cat gpurun_out/r2e_$N.txt
  ( time oracle/_ref/trust4 -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s${N}_1.fq -2 $D/s${N}_2.fq -o $D/ref$N ) 2>&1 | tail -4
  cmp $D/mine${N}_raw.out $D/ref${N}_raw.out && cmp $D/mine${N}_assembled_reads.fa $D/ref${N}_assembled_reads.fa && echo IDENTICAL_$N
done
