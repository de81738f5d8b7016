# which change moved the rough-annotation kernels: the same 600 k reads through the library built at four commits
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests")
import t4libs
np.save("/tmp/ab_reads.npy", t4libs.Synth(6000, 1).next_reads(300000))
PY
for v in c0 vC vA vB vE; do
  if [ $v = fix ]; then L=$GRAFT_REPO_ROOT/trust4_amd/libt4hip.so; else L=$GRAFT_REPO_ROOT/trust4_amd/variants/$v/libt4hip.so; fi
  T4_LIB=$L timeout 20 python tools/annot_ab.py /tmp/ab_reads.npy $v 2>&1 | tail -1 | tee -a gpurun_out/r2z_annot_ab.txt
done
