#!/bin/bash
# developer tool: A/B of the training benchmark on ONE box (boxes of the pool differ by +-2 %):  bash tools/ab_bench.sh VAR=1 [VAR2=1 ...]
# runs bench.py without and with each given environment setting, alternating, two rounds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
one() { env "$@" timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms  rollouts %.1f  update %.1f' % (d['ms_per_step'], d['phases_ms_per_step']['collect+det_rollout'], d['phases_ms_per_step']['update']))"; }
for round in 1 2; do
  echo "base     : $(one DGPPO_AB_DUMMY=1)"
  for v in "$@"; do echo "$v : $(one $v)"; done
done
