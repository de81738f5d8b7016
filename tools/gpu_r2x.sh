# GPU parity suite on the current build, then A/B of the run-time options at 100 k pairs
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r2x_gpu_tests.txt; cat gpurun_out/r2x_gpu_tests.txt
export T4_TIMING=1
D=/tmp/r2x; mkdir -p $D
zcat data/hg38_bcrtcr.fa.gz > $D/ref.fa
run() { # tag N [env...]
  local tag=$1 N=$2; shift 2
  ( time env "$@" trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s${N}_1.fq -2 $D/s${N}_2.fq -o $D/v$tag ) 2>&1 | grep "timing: AddRead query path\|timing: assembler host\|real\|phase \|GPU query rounds" > gpurun_out/r2x_${tag}_$N.txt
  echo "== $tag $N"; grep "real\|first launch" gpurun_out/r2x_${tag}_$N.txt; md5sum $D/v${tag}_raw.out | cut -c1-32
}
N=100000
tools/t4synth data/hg38_bcrtcr.fa.gz $N $((N/50)) 1 $D/s$N > /dev/null
run new $N
run heavy4 $N T4_HEAVY_AHEAD=4
run heavy8 $N T4_HEAVY_AHEAD=8
run spin $N T4_SCHEDULE_SPIN=1
