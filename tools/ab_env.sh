# A/B of run-time switches of the default build: every argument is one environment setting ("A=1 B=2"), "-" = none
mkdir -p gpurun_out; : > gpurun_out/ab_env.txt
N=${N:-2000000}
for E in "$@"; do
  [ "$E" = "-" ] && E=""
  echo "[$E] $(env $E python tools/gpu_pass.py $N 3 2>&1 | tail -1)" | tee -a gpurun_out/ab_env.txt
done
