cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
S=gpurun_out/r2_run3_status.log; rm -f $S
run() { name=$1; to=$2; shift 2
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1; rc=$?
  echo "$name rc=$rc" | tee -a $S
  if [ $rc -ge 124 ] && [ $rc -le 137 ]; then echo "timeout/kill: stopping" | tee -a $S; exit $rc; fi
}
run r2c_t_env 600 python -m pytest tests/test_env_gpu.py -q -m gpu
run r2c_stamps 300 env DGPPO_HIP_LIB=$GRAFT_REPO_ROOT/dgppo_amd/csrc/libdgppo_hip_stamps.so python tools/stamps_wave.py
cat gpurun_out/r2c_stamps.log; tail -n 3 gpurun_out/r2c_t_env.log
