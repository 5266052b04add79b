#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_nn_gpu.py -x -q -m gpu -k "dense" > $O/r3c_t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/r3c_t.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 tools/bench_nn.py dense > $O/r3c_bench_nn.log 2>&1; echo "bench_nn rc=$?"; grep "bwd_w.*K=8 " $O/r3c_bench_nn.log
timeout -k 10 300 env DGPPO_DENSE_NO_SMALLK=1 python3 tools/bench_nn.py dense > $O/r3c_bench_nn_off.log 2>&1; echo "bench_nn rc=$?"; grep "K=8 " $O/r3c_bench_nn_off.log
for cfg in "A=1" "DGPPO_DENSE_NO_SMALLK_BWD=1" "A=1" "DGPPO_DENSE_NO_SMALLK_BWD=1"; do
timeout -k 10 300 env $cfg python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > $O/r3c_bench_x.log 2>&1
python3 - "$cfg" <<'PY'
import json,sys
f="gpurun_out/r3c_bench_x.log"
l=[x for x in open(f) if x.startswith("{")]
d=json.loads(l[-1]); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],1), d["phases_ms_per_step"])
PY
done
