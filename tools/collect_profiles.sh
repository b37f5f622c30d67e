#!/bin/bash
# Round artifacts on the GPU box (run through gpurun from the repository root): the default bench line (configs[2]),
# the rocprofv3 kernel trace of the same command, the two PMC passes (each in a run of its own), the FETCH_SIZE
# calibration for this library's access widths, and the other workloads' lines.  usage: tools/collect_profiles.sh <tag> [part]
set -e
TAG=$1
PART=${2:-all}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ $PART = all ] || [ $PART = bench ]; then
timeout -k 10 600 python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
timeout -k 10 400 python3 $R/bench.py --workload load > $O/${TAG}_bench_load.json 2> $O/${TAG}_bench_load.err
fi
if [ $PART = all ] || [ $PART = prof ]; then
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof_${TAG} -o ${TAG} -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $O/${TAG}_prof.json 2> $O/${TAG}_prof.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_load -o ${TAG}_load -- python3 $R/bench.py --workload load --steps 3 --warmup 1 --no-cpu > $O/${TAG}_prof_load.json 2> $O/${TAG}_prof_load.err
fi
if [ $PART = all ] || [ $PART = pmc ]; then
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_${TAG} -o f -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $O/${TAG}_pmc_f.json 2> $O/${TAG}_pmc_f.err
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_${TAG} -o w -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $O/${TAG}_pmc_w.json 2> $O/${TAG}_pmc_w.err
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_${TAG}_load -o f -- python3 $R/bench.py --workload load --steps 1 --warmup 0 --no-cpu > $O/${TAG}_pmc_fl.json 2> $O/${TAG}_pmc_fl.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_${TAG}_load -o w -- python3 $R/bench.py --workload load --steps 1 --warmup 0 --no-cpu > $O/${TAG}_pmc_wl.json 2> $O/${TAG}_pmc_wl.err
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_calib_${TAG} -o c -- $R/tools/dev/fetch_calib > $O/${TAG}_calib.txt 2> $O/${TAG}_calib.err
fi
if [ $PART = all ] || [ $PART = other ]; then
timeout -k 10 600 python3 $R/bench.py --workload bigdoc > $O/${TAG}_bench_bigdoc.json 2> $O/${TAG}_bench_bigdoc.err
timeout -k 10 300 python3 $R/bench.py --workload sync --steps 3 --warmup 1 > $O/${TAG}_bench_sync.json 2> $O/${TAG}_bench_sync.err
timeout -k 10 300 python3 $R/bench.py --workload apply --steps 3 --warmup 1 --no-cpu > $O/${TAG}_bench_apply.json 2> $O/${TAG}_bench_apply.err
timeout -k 10 300 python3 $R/bench.py --workload trace --steps 20 --warmup 3 > $O/${TAG}_bench_trace.json 2> $O/${TAG}_bench_trace.err
fi
ls $O | grep ${TAG} | head -40
