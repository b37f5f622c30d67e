#!/bin/bash
# Round artifacts on the GPU box (run through gpurun from the repository root): the bench line, the rocprofv3 kernel
# trace of the same command and the two PMC passes (each in a run of its own).  usage: tools/collect_profiles.sh <tag>
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_${TAG} -o ${TAG} -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $O/${TAG}_prof.json 2> $O/${TAG}_prof.err
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_${TAG} -o f -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $O/${TAG}_pmc_f.json 2> $O/${TAG}_pmc_f.err
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_${TAG} -o w -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $O/${TAG}_pmc_w.json 2> $O/${TAG}_pmc_w.err
timeout -k 10 400 python3 $R/bench.py --workload merge --steps 2 --warmup 1 > $O/${TAG}_bench_merge.json 2> $O/${TAG}_bench_merge.err
timeout -k 10 300 python3 $R/bench.py --workload sync --steps 3 --warmup 1 > $O/${TAG}_bench_sync.json 2> $O/${TAG}_bench_sync.err
timeout -k 10 300 python3 $R/bench.py --workload apply --steps 3 --warmup 1 --no-cpu > $O/${TAG}_bench_apply.json 2> $O/${TAG}_bench_apply.err
ls -la $O/prof_${TAG} $O/pmc_fetch_${TAG} $O/pmc_write_${TAG}
