# A/B of kernel variants on one GPU box: trust4_amd/variants/libt4hip_*.so (built with -DT4_OPT_x=0 etc.) against the default
# build, same bench command (tools/quick_bench.sh prints reads/s, kernel ms, reads per tier). Outputs under gpurun_out/.
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
echo "default: $(bash tools/quick_bench.sh)" | tee gpurun_out/ab.txt
for L in trust4_amd/variants/libt4hip_*.so; do
  case $L in *phases*) continue;; esac
  echo "$(basename $L): $(T4_LIB=$PWD/$L bash tools/quick_bench.sh)" | tee -a gpurun_out/ab.txt
done
echo "default again: $(bash tools/quick_bench.sh)" | tee -a gpurun_out/ab.txt
[ -f trust4_amd/variants/libt4hip_phases.so ] && python tools/gpu_phases.py $PWD/trust4_amd/variants/libt4hip_phases.so 400000 > gpurun_out/phases.txt 2>&1
tail -20 gpurun_out/phases.txt
