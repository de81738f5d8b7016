# config C2 itself (1 M pairs, 20 k clones, seed 1) through whole stage 1; the reference's outputs of the same files were taken
# in session r2i (profiles/r02_c2_reference_log.txt): md5 of its _raw.out = 17170ea86b87c3b2940d7e5b17382469
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
export T4_TIMING=1
D=/tmp/r2c2; mkdir -p $D
zcat data/hg38_bcrtcr.fa.gz > $D/ref.fa
tools/t4synth data/hg38_bcrtcr.fa.gz 1000000 20000 1 $D/c2 > /dev/null
( time T4_STATS_JSON=$R/gpurun_out/r02_c2_stats.json trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/c2_1.fq -2 $D/c2_2.fq -o $D/mine ) > gpurun_out/r02_c2_full.txt 2>&1
grep -v "Processed\|Read in and count" gpurun_out/r02_c2_full.txt | tail -12
md5sum $D/mine_raw.out $D/mine_assembled_reads.fa | tee gpurun_out/r02_c2_md5.txt
