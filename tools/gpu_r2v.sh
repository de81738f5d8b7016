mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
export T4_TIMING=1
D=/tmp/r2v; mkdir -p $D
zcat data/hg38_bcrtcr.fa.gz > $D/ref.fa
N=300000
tools/t4synth data/hg38_bcrtcr.fa.gz $N $((N/50)) 1 $D/s$N > /dev/null
for V in new old new old; do
if [ $V = old ]; then export LD_LIBRARY_PATH=$R/trust4_amd/variants/noreg64; else unset LD_LIBRARY_PATH; fi
( time trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s${N}_1.fq -2 $D/s${N}_2.fq -o $D/v$V ) 2>&1 | grep "timing: AddRead query path host\|real" > gpurun_out/r2v_$V.txt
echo "$V"; cat gpurun_out/r2v_$V.txt; md5sum $D/v${V}_raw.out
done
