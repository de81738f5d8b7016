mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
export T4_TIMING=1
D=/tmp/r2r; mkdir -p $D
zcat data/hg38_bcrtcr.fa.gz > $D/ref.fa
N=300000
tools/t4synth data/hg38_bcrtcr.fa.gz $N $((N/50)) 1 $D/s$N > /dev/null
for V in 6 12 24 64; do
( time T4_QUERY_AHEAD=$V trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s${N}_1.fq -2 $D/s${N}_2.fq -o $D/v$V ) 2>&1 | grep "timing: AddRead query path host\|real\|Finish assembly" | sed 's/; [0-9]* image deltas.*//' > gpurun_out/r2r_$V.txt
echo "ahead $V"; cat gpurun_out/r2r_$V.txt
done
