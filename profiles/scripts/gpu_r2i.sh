mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
export T4_TIMING=1
D=/tmp/r2i; mkdir -p $D
zcat data/hg38_bcrtcr.fa.gz > $D/ref.fa
N=1000000
tools/t4synth data/hg38_bcrtcr.fa.gz $N 20000 1 $D/c2 > /dev/null
( ( time oracle/_ref/trust4 -t 16 --skipMateExtension -f $D/ref.fa -1 $D/c2_1.fq -2 $D/c2_2.fq -o $D/ref ) > gpurun_out/r2i_c2_ref.txt 2>&1 ) &
( time T4_STATS_JSON=$R/gpurun_out/r2i_c2_stats.json trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/c2_1.fq -2 $D/c2_2.fq -o $D/mine ) > gpurun_out/r2i_c2_full.txt 2>&1
grep "timing\|real\|Finish assembly\|failed\|exceeds\|fault\|Memory" gpurun_out/r2i_c2_full.txt | grep -v "Processed"; tail -4 gpurun_out/r2i_c2_full.txt
wait
tail -6 gpurun_out/r2i_c2_ref.txt
( cmp $D/mine_raw.out $D/ref_raw.out && cmp $D/mine_assembled_reads.fa $D/ref_assembled_reads.fa && echo IDENTICAL_C2 ) | tee gpurun_out/r2i_c2_identical.txt
md5sum $D/mine_raw.out $D/ref_raw.out | tee -a gpurun_out/r2i_c2_identical.txt
