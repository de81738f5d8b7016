mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
export T4_TIMING=1
D=/tmp/r2l; mkdir -p $D
zcat data/hg38_bcrtcr.fa.gz > $D/ref.fa
for N in 100000 300000; do
tools/t4synth data/hg38_bcrtcr.fa.gz $N $((N/50)) 1 $D/s$N > /dev/null
( time T4_STATS_JSON=$R/gpurun_out/r2l_stats_$N.json trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s${N}_1.fq -2 $D/s${N}_2.fq -o $D/mine$N ) 2>&1 | grep "timing: A\|timing: ass\|real\|Finish assembly" > gpurun_out/r2l_$N.txt
cat gpurun_out/r2l_$N.txt
md5sum $D/mine${N}_raw.out $D/mine${N}_assembled_reads.fa
done
