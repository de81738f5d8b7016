#!/bin/bash
# round 5, session o: the library as committed at the end (export / merge buffers behind guards): the counter tests and smoke()
# gpurun --timeout 60 -- 'bash profiles/scripts/gpu_r5o.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5o; mkdir -p $O
export TMPDIR=/tmp
timeout 25 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt | cut -c1-200
timeout 30 python -m pytest tests/test_zz_kmer_count_gpu.py -m gpu -q -x -k "export_merge or counts_and_stats" > $O/gpu_tests_kc.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_kc.txt; tail -3 $O/gpu_tests_kc.txt | cut -c1-200
echo "elapsed $SECONDS"
