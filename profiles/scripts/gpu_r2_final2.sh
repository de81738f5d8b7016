# end of round 2: smoke, GPU parity suite, the bench line, then 300 k pairs and a per-phase profile of the final build
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > gpurun_out/r02_smoke.txt; cat gpurun_out/r02_smoke.txt
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r02_gpu_tests.txt; cat gpurun_out/r02_gpu_tests.txt
timeout 600 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; head -c 600 gpurun_out/r02_bench.json; echo
export T4_TIMING=1
D=/tmp/r2y; mkdir -p $D
zcat data/hg38_bcrtcr.fa.gz > $D/ref.fa
run() { # tag N [env...]
  local tag=$1 N=$2; shift 2
  ( time env "$@" trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/s${N}_1.fq -2 $D/s${N}_2.fq -o $D/v$tag ) 2>&1 | grep "timing: AddRead query path\|timing: assembler host\|real\|phase \|GPU query rounds" > gpurun_out/r2y_${tag}_$N.txt
  echo "== $tag $N"; grep "real\|first launch" gpurun_out/r2y_${tag}_$N.txt; md5sum $D/v${tag}_raw.out | cut -c1-32
}
N=100000
tools/t4synth data/hg38_bcrtcr.fa.gz $N $((N/50)) 1 $D/s$N > /dev/null
run phase $N LD_LIBRARY_PATH=$R/trust4_amd/variants/phase T4_PHASE_TIMING=1 T4_PHASE_DUMP=1
N=300000
tools/t4synth data/hg38_bcrtcr.fa.gz $N $((N/50)) 1 $D/s$N > /dev/null
run new $N
