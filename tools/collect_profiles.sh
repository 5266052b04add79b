#!/bin/bash
# Collects the rocprofv3 artefacts of a round on the GPU box (run from the repo root through gpurun):
#   bash tools/collect_profiles.sh r02
# Everything lands under gpurun_out/profiles_$TAG/; copy the summaries into profiles/ afterwards (tools/profiles_import.py).
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/profiles_$TAG
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; to=$2; shift 2
  timeout -k 10 $to "$@" > $OUT/$name.log 2>&1; rc=$?
  echo "$name rc=$rc" | tee -a $OUT/status.log
  if [ $rc -ge 124 ] && [ $rc -le 137 ]; then echo "timeout/kill: stopping" | tee -a $OUT/status.log; exit $rc; fi
}
# the roofline kernel alone: kernel trace + stats, then the two HBM counters in separate passes (guide: FETCH/WRITE do not fit one pass)
run env_kt 300 env SIZES=4096,16384 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/env_kt -- python3 tools/bench_env.py
run env_fetch 300 env SIZES=4096 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/env_fetch -- python3 tools/bench_env.py
run env_write 300 env SIZES=4096 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/env_write -- python3 tools/bench_env.py
run env_insts 300 env SIZES=4096 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/env_insts -- python3 tools/bench_env.py
# the whole benchmark and the update phase alone
run bench_kt 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_kt -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline
run update_kt 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/update_kt -- python3 tools/prof_update_only.py
# un-profiled reference numbers of the same build
run bench_plain 600 python3 bench.py
run stamps 300 env DGPPO_HIP_LIB=$GRAFT_REPO_ROOT/dgppo_amd/csrc/libdgppo_hip_stamps.so python3 tools/stamps_wave.py
cat $OUT/status.log
