#!/bin/bash
# Collects the rocprofv3 artefacts of a round on the GPU box (run from the repo root through gpurun):
#   bash tools/collect_profiles.sh r03
# Everything lands under gpurun_out/profiles_$TAG/; copy the summaries into profiles/ afterwards (tools/profiles_import.py).
# Counter passes are separate runs with --pmc only (never combined with tracing); the program comes directly after `--`.
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/profiles_$TAG
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; to=$2; shift 2
  timeout -k 10 $to "$@" > $OUT/$name.log 2>&1; rc=$?
  echo "$name rc=$rc" | tee -a $OUT/status.log
  if [ $rc -ge 124 ] && [ $rc -le 137 ]; then echo "timeout/kill: stopping" | tee -a $OUT/status.log; exit $rc; fi
}
# ---- the roofline kernel alone: kernel trace + stats, then the two HBM counters in separate passes (FETCH/WRITE do not fit one pass)
export SIZES=4096,16384
run env_kt 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/env_kt -- python3 tools/bench_env.py
export SIZES=4096
run env_fetch 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/env_fetch -- python3 tools/bench_env.py
run env_write 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/env_write -- python3 tools/bench_env.py
export SIZES=4096,16384
run env_insts 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/env_insts -- python3 tools/bench_env.py
run env_insts2 300 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/env_insts2 -- python3 tools/bench_env.py
unset SIZES
# ---- the update phase (the real kernels at the real sizes; one repetition, single stream): trace, then four counter passes
export MS=0 REPS=1
run update_kt 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/update_kt -- python3 tools/prof_update_only.py
run nn_fetch 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/nn_fetch -- python3 tools/prof_update_only.py
run nn_write 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/nn_write -- python3 tools/prof_update_only.py
run nn_mfma 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/nn_mfma -- python3 tools/prof_update_only.py
run nn_lds 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/nn_lds -- python3 tools/prof_update_only.py
unset MS REPS
# ---- the whole benchmark under the tracer, then un-profiled reference numbers of the same build
run bench_kt 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_kt -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline
run bench_plain 900 python3 bench.py
export DGPPO_HIP_LIB=$GRAFT_REPO_ROOT/dgppo_amd/csrc/libdgppo_hip_stamps.so
run stamps 300 python3 tools/stamps_wave.py
unset DGPPO_HIP_LIB
run valu_rate 200 ./tools/micro/valu_rate
cat $OUT/status.log
