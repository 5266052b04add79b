#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_env_gpu.py -x -q -m gpu > $O/r3f_t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -25 $O/r3f_t.log | cut -c1-250
