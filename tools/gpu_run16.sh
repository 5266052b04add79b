#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 1100 python3 -m pytest tests/test_engine_gpu.py tests/test_api_gpu.py -x -q -m gpu > $O/r3g_t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -30 $O/r3g_t.log | cut -c1-250
