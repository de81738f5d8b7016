mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
export T4_TIMING=1
D=/tmp/r2m; mkdir -p $D
zcat data/hg38_bcrtcr.fa.gz > $D/ref.fa
N=300000
tools/t4synth data/hg38_bcrtcr.fa.gz $N $((N/50)) 1 $D/s$N > /dev/null
( time LD_LIBRARY_PATH=$R/trust4_amd/variants/phase T4_PHASE_DUMP=1 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s${N}_1.fq -2 $D/s${N}_2.fq -o $D/mine ) 2>&1 | grep "phase \|Finish assembly\|real\|timing: A" > gpurun_out/r2m_phases.txt
cat gpurun_out/r2m_phases.txt
