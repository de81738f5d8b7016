#!/bin/bash
# round 4, session o: the whole GPU suite and smoke() on HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4o; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt; tail -6 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt | cut -c1-200
