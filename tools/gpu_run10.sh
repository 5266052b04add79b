cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/kt_upd
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_upd -- python3 tools/prof_update_only.py > gpurun_out/r2_prof_update.log 2>&1; echo rc=$?
tail -5 gpurun_out/r2_prof_update.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bench_b.log 2>gpurun_out/r2_bench_b.err; echo rc=$?; tail -c 1500 gpurun_out/r2_bench_b.log
