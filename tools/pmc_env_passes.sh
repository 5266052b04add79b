cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/pmc_env; rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $O/avail.txt | sort -u > $O/sq_names.txt; wc -l $O/sq_names.txt
export SIZES=16384
timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/p1 -- python3 tools/bench_env.py > $O/p1.log 2>&1; echo p1 rc=$?
timeout -k 5 200 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p2 -- python3 tools/bench_env.py > $O/p2.log 2>&1; echo p2 rc=$?
timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/p3 -- python3 tools/bench_env.py > $O/p3.log 2>&1; echo p3 rc=$?
ls $O/p1 $O/p2 $O/p3 | head; find $O -name "*counter_collection.csv" | head
