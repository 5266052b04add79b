cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
S=gpurun_out/r2_run12_status.log; rm -f $S
run() { name=$1; to=$2; shift 2
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1; rc=$?
  echo "$name rc=$rc" | tee -a $S
  if [ $rc -ge 124 ] && [ $rc -le 137 ]; then echo "timeout/kill: stopping" | tee -a $S; exit $rc; fi
}
run r2j_t_eng 900 python -m pytest tests/test_engine_gpu.py tests/test_api_gpu.py -q -m gpu -x
run r2j_bench 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline
run r2j_smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
tail -n 15 gpurun_out/r2j_t_eng.log; tail -n 2 gpurun_out/r2j_smoke.log
python -c "import json; d=json.loads(open('gpurun_out/r2j_bench.log').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['phases_ms_per_step'], d['update_host_ms_last_step'], d['mfma']['util'])"
