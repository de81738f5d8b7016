#!/bin/bash
# round 4, session q: --readShard 4 ranks on the box's one GPU (file transport), outputs against the single-process run
mkdir -p gpurun_out/r4q
timeout 140 python -m pytest tests/test_dist_gloo.py -m gpu -k read_shard -x -q > gpurun_out/r4q/pytest.txt 2>&1
echo "exit $?" >> gpurun_out/r4q/pytest.txt
tail -5 gpurun_out/r4q/pytest.txt
