# SQ counters of the query kernels (one pass set per rocprofv3 run; 8 SQ slots): issue/wait split of the wave cycles.
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/sq1 -o sq -- python $R/tools/gpu_pass.py 400000 1 > $R/gpurun_out/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/sq2 -o sq -- python $R/tools/gpu_pass.py 400000 1 > $R/gpurun_out/sq2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("sq1", "sq2"):
    f = glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True)
    if not f: print(d, "no csv"); print(open("gpurun_out/%s.log" % d).read()[-800:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if "queryKernel" in k: print(d, k, dict(v))
PY
