#!/bin/bash
# Collect the judged profiles of the headline bench on the GPU box (run from the repo
# root through gpurun): bench JSON, rocprofv3 kernel stats, and the two PMC passes
# (FETCH_SIZE, WRITE_SIZE; counters in their own runs with --kernel-trace only).
#   bash profiles/collect.sh [tag]      -> gpurun_out/<tag>/
TAG=${1:-final}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --no-cpu --steps 50 --warmup 10 --profile-steps 0 > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $ROOT/bench.py --no-cpu --steps 5 --warmup 2 --profile-steps 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $ROOT/bench.py --no-cpu --steps 5 --warmup 2 --profile-steps 0 > $OUT/pmc_write.log 2>&1
python3 $ROOT/profiles/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/traffic.json
python3 $ROOT/profiles/pipe_roles.py $OUT/stats/bench_kernel_trace.csv > $OUT/pipe_roles.txt 2>&1
tail -1 $OUT/bench.json
