#!/bin/bash
# SQ counter pass over source loads of the headline workload (development aid): usage tools/dev/pmc_sq_text.sh <tag> [docs]
TAG=$1; DOCS=${2:-10000}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_sq_$TAG -o sq -- python3 $R/tools/dev/load_text_time.py $DOCS 1 0 > /dev/null 2> $R/gpurun_out/${TAG}_pmc_sq.err
cd $R && python3 - <<PY
import csv,glob
from collections import defaultdict
f=glob.glob('gpurun_out/pmc_sq_$TAG/*counter_collection.csv')[0]
tot=defaultdict(lambda: defaultdict(float)); n=defaultdict(int)
for r in csv.DictReader(open(f)):
    tot[r['Kernel_Name']][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVES': n[r['Kernel_Name']]+=1
for k,v in sorted(tot.items(), key=lambda kv:-kv[1].get('SQ_WAVE_CYCLES',0))[:14]:
    l=n[k] or 1
    print(k[:44], 'launches',l, ' '.join(f"{c[3:]}={x/l:.3g}" for c,x in sorted(v.items())))
PY
