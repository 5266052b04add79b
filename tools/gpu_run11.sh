cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
S=gpurun_out/r2_run11_status.log; rm -f $S
run() { name=$1; to=$2; shift 2
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1; rc=$?
  echo "$name rc=$rc" | tee -a $S
  if [ $rc -ge 124 ] && [ $rc -le 137 ]; then echo "timeout/kill: stopping" | tee -a $S; exit $rc; fi
}
run r2i_t_nn 900 python -m pytest tests/test_nn_gpu.py tests/test_engine_gpu.py tests/test_golden.py -q -m gpu
run r2i_upd 300 python tools/prof_update_only.py
run r2i_bench 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline
tail -n 5 gpurun_out/r2i_t_nn.log; grep rep gpurun_out/r2i_upd.log; tail -c 900 gpurun_out/r2i_bench.log
