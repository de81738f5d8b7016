# round 2: the measurement session. tests, the bench line, rocprofv3 kernel stats of the bench step's command, PMC passes.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r02_gpu_tests.txt; cat gpurun_out/r02_gpu_tests.txt
timeout 1500 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; tail -c 1500 gpurun_out/r02_bench.json; echo
# the step's command under rocprofv3 (same batch recipe: C2 at 0.1, seed 1)
D=/tmp/r2p; mkdir -p $D
zcat data/hg38_bcrtcr.fa.gz > $D/ref.fa
tools/t4synth data/hg38_bcrtcr.fa.gz 100000 2000 1 $D/b1 > /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_step -o r02 -- $R/trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/b1_1.fq -2 $D/b1_2.fq -o $D/prof > /dev/null 2>&1
find $R/gpurun_out/prof_step -name "*kernel_stats*" | head -1 | xargs -I{} cp {} $R/gpurun_out/r02_step_kernel_stats.csv
head -12 $R/gpurun_out/r02_step_kernel_stats.csv
# HBM-side traffic of the step's command (separate passes per counter)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_$C -o pmc -- $R/trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $D/ref.fa -1 $D/b1_1.fq -2 $D/b1_2.fq -o $D/prof > /dev/null 2>&1
  F=$(find $R/gpurun_out/pmcs_$C -name "*counter_collection.csv" | head -1)
  python - "$F" $C > $R/gpurun_out/r02_step_pmc_$C.txt <<'PY'
import csv, re, sys
from collections import defaultdict
acc, calls = defaultdict(float), defaultdict(int)
for row in csv.DictReader(open(sys.argv[1])):
    if row.get("Counter_Name") != sys.argv[2]: continue
    name = re.sub(r"\(.*", "", row["Kernel_Name"]).strip()
    acc[name] += float(row["Counter_Value"]); calls[name] += 1
for k in sorted(acc, key=lambda k: -acc[k]): print("%s\t%d launches\t%.0f %s units (KB)\t%.3f GB" % (k, calls[k], acc[k], sys.argv[2], acc[k] * 1024 / 1e9))
PY
  cat $R/gpurun_out/r02_step_pmc_$C.txt | head -4
  rm -rf $R/gpurun_out/pmcs_$C
done
# the data-parallel pass alone (C2 batch resident, two passes), kernel stats + PMC
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ann -o r02 -- python $R/tools/gpu_pass.py 2000000 3 > $R/gpurun_out/r02_annotate_pass.txt 2>&1
find $R/gpurun_out/prof_ann -name "*kernel_stats*" | head -1 | xargs -I{} cp {} $R/gpurun_out/r02_annotate_kernel_stats.csv
head -9 $R/gpurun_out/r02_annotate_kernel_stats.csv; tail -1 $R/gpurun_out/r02_annotate_pass.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -o pmc -- python $R/tools/gpu_pass.py 2000000 1 > /dev/null 2>&1
  F=$(find $R/gpurun_out/pmc_$C -name "*counter_collection.csv" | head -1); cp $F $R/gpurun_out/r02_pmc_$C.csv
done
python $R/tools/pmc_summary.py $R/gpurun_out/r02_pmc_FETCH_SIZE.csv $R/gpurun_out/r02_pmc_WRITE_SIZE.csv $R/gpurun_out/r02_pmc_summary.json
python -c "import json; d=json.load(open('$R/gpurun_out/r02_pmc_summary.json')); print({k: d[k] for k in ('query_kernels_fetch_bytes_raw','query_kernels_write_bytes_raw','traffic_bytes')})"
rm -rf $R/gpurun_out/prof_step $R/gpurun_out/prof_ann $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
