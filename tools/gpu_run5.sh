cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
S=gpurun_out/r2_run5_status.log; rm -f $S
run() { name=$1; to=$2; shift 2
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1; rc=$?
  echo "$name rc=$rc" | tee -a $S
  if [ $rc -ge 124 ] && [ $rc -le 137 ]; then echo "timeout/kill: stopping" | tee -a $S; exit $rc; fi
}
run r2e_t_env 600 python -m pytest tests/test_env_gpu.py -q -m gpu
run r2e_bench_env 300 env SIZES=1,4096,8192,16384,65536 python tools/bench_env.py
run r2e_bench_env_w6 300 env DGPPO_HIP_LIB=$GRAFT_REPO_ROOT/dgppo_amd/csrc/libdgppo_hip_w6.so SIZES=4096,16384,65536 python tools/bench_env.py
run r2e_stamps 300 env DGPPO_HIP_LIB=$GRAFT_REPO_ROOT/dgppo_amd/csrc/libdgppo_hip_stamps.so python tools/stamps_wave.py
tail -n 3 gpurun_out/r2e_t_env.log; cat gpurun_out/r2e_bench_env.log gpurun_out/r2e_bench_env_w6.log gpurun_out/r2e_stamps.log
