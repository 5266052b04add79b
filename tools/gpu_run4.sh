cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
S=gpurun_out/r2_run4_status.log; rm -f $S
run() { name=$1; to=$2; shift 2
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1; rc=$?
  echo "$name rc=$rc" | tee -a $S
  if [ $rc -ge 124 ] && [ $rc -le 137 ]; then echo "timeout/kill: stopping" | tee -a $S; exit $rc; fi
}
run r2d_t_env 600 python -m pytest tests/test_env_gpu.py -q -m gpu
run r2d_bench_env 300 env SIZES=1,4096,8192,16384,65536 python tools/bench_env.py
run r2d_stamps 300 env DGPPO_HIP_LIB=$GRAFT_REPO_ROOT/dgppo_amd/csrc/libdgppo_hip_stamps.so python tools/stamps_wave.py
rm -rf gpurun_out/pmc_w
run r2d_pmc 300 env SIZES=4096 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc_w -- python3 tools/bench_env.py
tail -n 3 gpurun_out/r2d_t_env.log; cat gpurun_out/r2d_bench_env.log gpurun_out/r2d_stamps.log
