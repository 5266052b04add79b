#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/r3h_t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 $O/r3h_t.log | cut -c1-250
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/r3h_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/r3h_smoke.log
timeout -k 10 600 python3 bench.py > $O/r3h_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/r3h_bench.log | cut -c1-1500
