cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu -x -k "rnn_option or epoch_ppo or policy_forward or Vl_forward or Vh_forward or update_targets or hip_graph or golden" > gpurun_out/r2l_t.log 2>&1; echo rc=$?
tail -n 40 gpurun_out/r2l_t.log
