cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
S=gpurun_out/r2_run8_status.log; rm -f $S
run() { name=$1; to=$2; shift 2
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1; rc=$?
  echo "$name rc=$rc" | tee -a $S
  if [ $rc -ge 124 ] && [ $rc -le 137 ]; then echo "timeout/kill: stopping" | tee -a $S; exit $rc; fi
}
run r2h_bench_env 300 env SIZES=1,1024,4096,8192,16384,65536 python tools/bench_env.py
rm -rf gpurun_out/kt
run r2h_kt 300 env SIZES=1,4096,16384 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 tools/bench_env.py
cut -c1-220 gpurun_out/r2h_bench_env.log; find gpurun_out/kt -name "*kernel_stats.csv" | head -1 | xargs head -8
