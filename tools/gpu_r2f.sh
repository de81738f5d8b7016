mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
export T4_TIMING=1
D=/tmp/r2f; mkdir -p $D
zcat data/hg38_bcrtcr.fa.gz > $D/ref.fa
N=100000
tools/t4synth data/hg38_bcrtcr.fa.gz $N $((N/50)) 1 $D/s$N > /dev/null
( time trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s${N}_1.fq -2 $D/s${N}_2.fq -o $D/mine$N ) 2>&1 | grep "timing\|real\|Finish assembly" > gpurun_out/r2f_$N.txt
cat gpurun_out/r2f_$N.txt
timeout 1500 python bench.py > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err; tail -c 6000 gpurun_out/r2f_bench.json; tail -5 gpurun_out/r2f_bench.err
