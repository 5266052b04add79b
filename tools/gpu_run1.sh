cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { # name, timeout, cmd...
  name=$1; to=$2; shift 2
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1; rc=$?
  echo "$name rc=$rc" | tee -a gpurun_out/r2_run1_status.log
  if [ $rc -ge 124 ] && [ $rc -le 137 ]; then echo "timeout/kill: stopping" | tee -a gpurun_out/r2_run1_status.log; exit $rc; fi
}
rm -f gpurun_out/r2_run1_status.log
run r2_t_engine 900 python -m pytest tests/test_engine_gpu.py -q -x -m gpu
run r2_t_rest 900 python -m pytest tests -q -m gpu --deselect tests/test_engine_gpu.py
run r2_bench_gloo2 600 env DGPPO_DIST_BACKEND=gloo python bench.py --gpus 2 --n-env 1024 --steps 2 --warmup 1 --no-cpu-baseline
run r2_bench1 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline
tail -3 gpurun_out/r2_t_engine.log gpurun_out/r2_t_rest.log
tail -1 gpurun_out/r2_bench_gloo2.log; tail -1 gpurun_out/r2_bench1.log
