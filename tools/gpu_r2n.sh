mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
export T4_TIMING=1
D=/tmp/r2n; mkdir -p $D
zcat data/hg38_bcrtcr.fa.gz > $D/ref.fa
N=300000
tools/t4synth data/hg38_bcrtcr.fa.gz $N $((N/50)) 1 $D/s$N > /dev/null
for T in 512 256; do
( time T4_AQ_THREADS=$T trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s${N}_1.fq -2 $D/s${N}_2.fq -o $D/mine$T ) 2>&1 | grep "timing: AddRead query path host\|real" > gpurun_out/r2n_$T.txt
echo "threads $T"; cat gpurun_out/r2n_$T.txt
done
cmp $D/mine512_raw.out $D/mine256_raw.out && echo same
