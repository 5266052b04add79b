#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_api_gpu.py -x -q -m gpu -k "test_py_cli or render" > $O/r3e_t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 $O/r3e_t.log | cut -c1-300
