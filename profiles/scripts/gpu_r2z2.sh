cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
D=/tmp/z2; mkdir -p $D; zcat data/hg38_bcrtcr.fa.gz > $D/ref.fa
tools/t4synth data/hg38_bcrtcr.fa.gz 20000 400 1 $D/s > /dev/null
for v in inl ni inl ni; do
  if [ $v = ni ]; then export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/trust4_amd/variants/ni; else unset LD_LIBRARY_PATH; fi
  T4_TIMING=1 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s_1.fq -2 $D/s_2.fq -o $D/o$v 2>&1 | grep "first launch" | sed "s/^/$v /" | tee -a gpurun_out/r2z2.txt
done
md5sum $D/oinl_raw.out $D/oni_raw.out | cut -c1-32
