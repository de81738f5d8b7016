# quick A/B on one GPU box: smoke of the default build, then tools/gpu_pass.py for the default and every trust4_amd/variants/libt4hip_*.so
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-80
N=${1:-1000000}
echo "default: $(python tools/gpu_pass.py $N 3 2>&1 | tail -1)" | tee gpurun_out/ab.txt
echo "default, T4_STATIC_STRIDE=1: $(T4_STATIC_STRIDE=1 python tools/gpu_pass.py $N 3 2>&1 | tail -1)" | tee -a gpurun_out/ab.txt
for L in trust4_amd/variants/libt4hip_*.so; do
  case $L in *phases*) continue;; esac
  echo "$(basename $L): $(T4_LIB=$PWD/$L python tools/gpu_pass.py $N 3 2>&1 | tail -1)" | tee -a gpurun_out/ab.txt
done
if [ -n "$2" ]; then python tools/gpu_phases.py $PWD/trust4_amd/variants/libt4hip_phases.so 400000 > gpurun_out/phases.txt 2>&1; tail -12 gpurun_out/phases.txt; fi
