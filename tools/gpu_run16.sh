#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_nn_gpu.py tests/test_engine_gpu.py tests/test_golden.py -x -q -m gpu > $O/r3l_t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/r3l_t.log | cut -c1-250
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > $O/r3l_bench.log 2>&1
python3 - <<'PY'
import json
l=[x for x in open("gpurun_out/r3l_bench.log") if x.startswith("{")]
d=json.loads(l[-1]); print(round(d["value"]), round(d["ms_per_step"],1), d["phases_ms_per_step"])
PY
done
