mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
export T4_TIMING=1
D=/tmp/r2g; mkdir -p $D
zcat data/hg38_bcrtcr.fa.gz > $D/ref.fa
for N in 100000 300000; do
tools/t4synth data/hg38_bcrtcr.fa.gz $N $((N/50)) 1 $D/s$N > /dev/null
( time T4_STATS_JSON=$R/gpurun_out/r2g_stats_$N.json trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s${N}_1.fq -2 $D/s${N}_2.fq -o $D/mine$N ) 2>&1 | grep "timing\|real\|Finish assembly\|Processed" > gpurun_out/r2g_$N.txt
cat gpurun_out/r2g_$N.txt
done
( time oracle/_ref/trust4 -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s300000_1.fq -2 $D/s300000_2.fq -o $D/ref300000 ) 2>&1 | tail -4
cmp $D/mine300000_raw.out $D/ref300000_raw.out && cmp $D/mine300000_assembled_reads.fa $D/ref300000_assembled_reads.fa && echo IDENTICAL_300000
