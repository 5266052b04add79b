#!/bin/bash
# developer tool: one rocprofv3 PMC pass over tools/bench_nn.py <which>   (bash tools/pmc_nn_pass.sh attn "SQ_WAVES SQ_INSTS_VALU ...")
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
W=${1:-attn}; shift
CTRS=${1:-"SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"}
TAG=${2:-p}
O=gpurun_out/pmc_nn_$TAG; rm -rf $O; mkdir -p $O
timeout -k 5 300 rocprofv3 --pmc $CTRS --output-format csv -d $O -- python3 tools/bench_nn.py $W > $O/log.txt 2>&1; echo "rc=$?"
python3 - "$O" <<'PY'
import csv, glob, collections, sys
for p in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name'].split('(')[0][-60:] + ' grid=' + r.get('Grid_Size', '?')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        n = len(next(iter(d.values())))
        print(k, 'launches', n)
        print('     ' + '  '.join('%s=%.3g' % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
