#!/bin/bash
# Training sanity with spread: the 400-iteration LidarSpread run for several seeds, for this build and for
# older builds checked out as worktrees (_r1/, _r2/: `git worktree add _r1 <round-1 commit>` + make).  Run on the GPU box:
#   bash tools/sanity_seeds.sh "0 1 2" ". _r2 _r1"
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
SEEDS=${1:-"0 1 2"}; TREES=${2:-". _r2 _r1"}
O=$GRAFT_REPO_ROOT/gpurun_out/sanity; rm -rf $O; mkdir -p $O
for tree in $TREES; do
  name=$(echo $tree | sed 's/^\.$/cur/; s/^_//')
  for s in $SEEDS; do
    ( cd $GRAFT_REPO_ROOT/$tree && timeout -k 10 400 python3 train.py --env LidarSpread -n 8 --obs 3 --algo dgppo --steps 400 --n-env-train 4096 \
        --n-env-test 256 --eval-interval 50 --seed $s --debug > $O/${name}_seed$s.log 2>&1 ); rc=$?
    echo "$name seed $s rc=$rc $(grep -c '^step' $O/${name}_seed$s.log) evals, last: $(grep '^step' $O/${name}_seed$s.log | tail -n 1)"
    if [ $rc -ge 124 ]; then echo "timed out: stopping"; exit $rc; fi
  done
done
