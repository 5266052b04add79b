cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
S=gpurun_out/r2_run13_status.log; rm -f $S
run() { name=$1; to=$2; shift 2
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1; rc=$?
  echo "$name rc=$rc" | tee -a $S
  if [ $rc -ge 124 ] && [ $rc -le 137 ]; then echo "timeout/kill: stopping" | tee -a $S; exit $rc; fi
}
run r2k_t 900 python -m pytest tests -q -m gpu -k "lagr or Vh_global or lagrangian or mean_agents"
tail -n 40 gpurun_out/r2k_t.log
