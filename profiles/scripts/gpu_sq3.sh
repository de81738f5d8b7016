mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/sq3 -o sq -- python $R/tools/gpu_pass.py 400000 1 > $R/gpurun_out/sq3.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/sq3/**/*counter_collection.csv", recursive=True)
if not f: print(open("gpurun_out/sq3.log").read()[-1500:])
else:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if "queryKernel" in k or "bin" in k: print(k, dict(v))
PY
