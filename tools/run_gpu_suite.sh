#!/bin/bash
# The round's standard GPU check (run from the repo root through gpurun):  bash tools/run_gpu_suite.sh [pytest -k expr]
# 1) the whole `-m gpu` suite in ONE process (log: gpurun_out/gpu_tests.log), 2) a short bench.py line.  A step that
# times out or is killed stops the script: no further GPU work after a hang.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
K=${1:-}
if [ -n "$K" ]; then
  timeout -k 10 1000 python3 -m pytest tests -q -m gpu -k "$K" -p no:cacheprovider > $O/gpu_tests.log 2>&1; rc=$?
else
  timeout -k 10 1000 python3 -m pytest tests -q -m gpu -p no:cacheprovider > $O/gpu_tests.log 2>&1; rc=$?
fi
echo "tests rc=$rc"; tail -25 $O/gpu_tests.log | cut -c1-300
if [ $rc -ge 124 ]; then echo "test run timed out / was killed: stopping"; exit $rc; fi
[ "${SKIP_BENCH:-0}" = "1" ] && exit $rc
timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_quick.json 2> $O/bench_quick.err; brc=$?
echo "bench rc=$brc"; python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_quick.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "phases_ms_per_step")}, d["roofline"]["us_per_launch"], d["roofline"]["frac"], d["roofline"]["at_4x_envs"])
except Exception as ex:
    print("no bench line:", ex)
PY
exit $rc
