mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
export T4_TIMING=1
D=/tmp/r2d; mkdir -p $D
zcat data/hg38_bcrtcr.fa.gz > $D/ref.fa
tools/t4synth data/hg38_bcrtcr.fa.gz 100000 2000 1 $D/s > /dev/null
( time trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s_1.fq -2 $D/s_2.fq -o $D/mine ) 2>&1 | tail -12 > gpurun_out/r2d_100k.txt
cat gpurun_out/r2d_100k.txt
( LD_LIBRARY_PATH=$R/trust4_amd/phase T4_PHASE_DUMP=1 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s_1.fq -2 $D/s_2.fq -o $D/mine2 ) 2>&1 | grep "phase \|Finish assembly" > gpurun_out/r2d_100k_phases.txt
cat gpurun_out/r2d_100k_phases.txt
cmp $D/mine_raw.out $D/mine2_raw.out && echo same
