mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
export T4_TIMING=1
D=/tmp/r2o; mkdir -p $D
zcat data/hg38_bcrtcr.fa.gz > $D/ref.fa
N=300000
tools/t4synth data/hg38_bcrtcr.fa.gz $N $((N/50)) 1 $D/s$N > /dev/null
( time trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s${N}_1.fq -2 $D/s${N}_2.fq -o $D/wide ) 2>&1 | grep "timing: AddRead query path host\|real" > gpurun_out/r2o_wide.txt
echo wide; cat gpurun_out/r2o_wide.txt
( time T4_AQ_INLINE_EXTEND=1 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s${N}_1.fq -2 $D/s${N}_2.fq -o $D/inl ) 2>&1 | grep "timing: AddRead query path host\|real" > gpurun_out/r2o_inline.txt
echo inline; cat gpurun_out/r2o_inline.txt
cmp $D/wide_raw.out $D/inl_raw.out && cmp $D/wide_assembled_reads.fa $D/inl_assembled_reads.fa && echo same
md5sum $D/wide_raw.out
