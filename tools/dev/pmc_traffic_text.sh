#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE: a pass each) of the kernels of one source load of the headline workload.
# usage: tools/dev/pmc_traffic_text.sh <tag> [docs]
TAG=$1; DOCS=${2:-10000}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_f_$TAG -o f -- python3 $R/tools/dev/load_text_time.py $DOCS 1 0 > /dev/null 2> $R/gpurun_out/${TAG}_pmc_f.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_w_$TAG -o w -- python3 $R/tools/dev/load_text_time.py $DOCS 1 0 > /dev/null 2> $R/gpurun_out/${TAG}_pmc_w.err || exit 1
cd $R && python3 tools/pmc_traffic.py text $(ls gpurun_out/pmc_f_$TAG/*counter_collection.csv | head -1) $(ls gpurun_out/pmc_w_$TAG/*counter_collection.csv | head -1) gpurun_out/traffic_text_$TAG.json && python3 - <<PY
import json
j=json.load(open('gpurun_out/traffic_text_$TAG.json'))
for k,v in sorted(j['kernels'].items(), key=lambda kv:-kv[1]['traffic'])[:16]:
    print(f"{k[:50]:50s} fetch {v['fetch']/1e9:7.2f} GB  write {v['write']/1e9:7.2f} GB  launches {v['launches']}")
PY
