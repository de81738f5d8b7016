# A/B of the heavy-read changes (key sort of overlaps, chunk-parallel pre-filter, lane-per-overlap walk, register sub-steps in the
# blocked sort) against the previous build, plus the 1024-thread workgroup option and a per-phase profile of the new build
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
export T4_TIMING=1
D=/tmp/r2w; mkdir -p $D
zcat data/hg38_bcrtcr.fa.gz > $D/ref.fa
run() { # tag N [env...]
  local tag=$1 N=$2; shift 2
  ( time env "$@" trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s${N}_1.fq -2 $D/s${N}_2.fq -o $D/v$tag ) 2>&1 | grep "timing: AddRead query path\|timing: assembler host\|real\|phase " > gpurun_out/r2w_${tag}_$N.txt
  echo "== $tag $N"; grep "real\|first launch" gpurun_out/r2w_${tag}_$N.txt; md5sum $D/v${tag}_raw.out | cut -c1-32
}
N=100000
tools/t4synth data/hg38_bcrtcr.fa.gz $N $((N/50)) 1 $D/s$N > /dev/null
run new $N
run old $N LD_LIBRARY_PATH=$R/trust4_amd/variants/r2old
run new1024 $N T4_AQ_THREADS=1024
run new $N
run phase $N LD_LIBRARY_PATH=$R/trust4_amd/variants/phase T4_PHASE_TIMING=1 T4_PHASE_DUMP=1
N=300000
tools/t4synth data/hg38_bcrtcr.fa.gz $N $((N/50)) 1 $D/s$N > /dev/null
run new $N
run old $N LD_LIBRARY_PATH=$R/trust4_amd/variants/r2old
