#!/bin/bash
# Collect the judged profiles of a bench workload on the GPU box (run from the repo
# root through gpurun): rocprofv3 kernel stats, the two PMC passes (FETCH_SIZE, WRITE_SIZE; counters
# in their own runs with --kernel-trace only) and the bench JSON line.
#   bash profiles/collect.sh [tag] [model: lsm|hdp|cc] [profile|bench|all]   -> gpurun_out/<tag>/
# `profile` comes first in a round's collection (collect_round.sh): the bench line's roofline divides
# by the rocprofv3 averages of THIS round (profiles/kernel_durations.json).
TAG=${1:-final}
MODEL=${2:-lsm}
WHAT=${3:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$WHAT" = "profile" ] || [ "$WHAT" = "all" ]; then
# (every profiler pass under a time limit of its own: a pass that hangs must not eat the call's budget.  The PMC
# passes serialise the dispatches of all queues; the HDP-LPCM loop's queue-level waits (hipStreamWaitValue32)
# never returned under them in round 5 - 40 GPU-minutes - so those passes use the gate kernel, as round 4 did)
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$MODEL -o bench -- python3 $ROOT/bench.py --model $MODEL --no-cpu --steps 50 --warmup 10 --profile-steps 0 > $OUT/stats_$MODEL.log 2>&1
DLSM_HDP_GATE=kernel timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$MODEL -o fetch -- python3 $ROOT/bench.py --model $MODEL --no-cpu --steps 5 --warmup 2 --profile-steps 0 > $OUT/pmc_fetch_$MODEL.log 2>&1
DLSM_HDP_GATE=kernel timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$MODEL -o write -- python3 $ROOT/bench.py --model $MODEL --no-cpu --steps 5 --warmup 2 --profile-steps 0 > $OUT/pmc_write_$MODEL.log 2>&1
python3 $ROOT/profiles/pmc_traffic.py $OUT/pmc_fetch_$MODEL $OUT/pmc_write_$MODEL > $OUT/traffic_$MODEL.json
python3 $ROOT/profiles/pipe_roles.py $OUT/stats_$MODEL/bench_kernel_trace.csv > $OUT/pipe_roles_$MODEL.txt 2>&1
find $OUT/stats_$MODEL -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$MODEL.csv
fi
if [ "$WHAT" = "bench" ] || [ "$WHAT" = "all" ]; then
python3 $ROOT/bench.py --model $MODEL > $OUT/bench_$MODEL.json 2> $OUT/bench_$MODEL.err
tail -1 $OUT/bench_$MODEL.json
fi
