cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu -x -k "rnn_option" > gpurun_out/r2m_t.log 2>&1; echo rc=$?
tail -n 40 gpurun_out/r2m_t.log
