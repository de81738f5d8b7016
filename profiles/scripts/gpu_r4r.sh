#!/bin/bash
# round 4, session r: RCCL bound by dlopen -- the two one-rank RCCL tests and the read-shard tests on the real engine
mkdir -p gpurun_out/r4r
timeout 160 python -m pytest tests/test_dist_gloo.py tests/test_stage1_e2e.py tests/test_abi.py -m gpu -k "read_shard or rccl" -x -q > gpurun_out/r4r/pytest.txt 2>&1
echo "exit $?" >> gpurun_out/r4r/pytest.txt
tail -5 gpurun_out/r4r/pytest.txt
