#!/bin/bash
# developer tool: time tuning variants of the wave-per-env kernel (make -C dgppo_amd/csrc variant V=... VFLAGS=...)
#   bash tools/bench_env_variants.sh w6u1 w5u1 ...      (run on the GPU box from the repo root)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in base "$@"; do
  lib=$GRAFT_REPO_ROOT/dgppo_amd/csrc/libdgppo_hip_$v.so
  [ "$v" = base ] && lib=$GRAFT_REPO_ROOT/dgppo_amd/csrc/libdgppo_hip.so
  DGPPO_HIP_LIB=$lib SIZES=${SIZES:-4096,16384} timeout -k 5 120 python3 tools/bench_env.py > gpurun_out/bev_$v.log 2>&1 || { echo "$v FAILED"; tail -n 5 gpurun_out/bev_$v.log; exit 1; }
  echo "== $v: $(grep parity gpurun_out/bev_$v.log)"
  grep -E "^[0-9]+ " gpurun_out/bev_$v.log | python3 -c "
import sys, json
for ln in sys.stdin:
    b, js = ln.split(' ', 1); d = json.loads(js)
    print('   B=%s api %.2f us (frac %.3f)  compact %.2f us' % (b, d['api']['us_per_launch'], d['api']['gbs'] / 8000, d['compact']['us_per_launch']))"
done
