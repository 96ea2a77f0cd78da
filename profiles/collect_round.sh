#!/bin/bash
# Everything profiles/README.md lists for a round, in one call on the GPU box (from the repo root,
# through gpurun; ~5 minutes).  Optional inputs under tmp_timing/ (untracked, built in the container first):
#   libtiming.so = the engine with -DDLSM_PIPE_TIMING (see profiles/pipe_timing.py),
#   valu_rates, sqrt_acc = hipcc -O3 --offload-arch=gfx950 of profiles/micro/valu_rates.hip, sqrt_acc.cpp
#   bash profiles/collect_round.sh <tag>      -> gpurun_out/<tag>/
TAG=${1:-round}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
# 1. the profiler's passes; 2. this round's stored durations and traffic (what the bench lines' rooflines divide by:
#    written where bench.py reads them, and into $OUT for the way home); 3. the bench lines
for m in lsm hdp cc ccu; do bash $ROOT/profiles/collect.sh $TAG $m profile > $OUT/collect_$m.log 2>&1; done
cd $ROOT
# the HDP-LPCM iteration launch by launch, on its two queues (the trace above) and on one (a trace of its own)
python3 profiles/iteration_timeline.py $OUT/stats_hdp/bench_kernel_trace.csv > $OUT/hdp_timeline_two_queues.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && DLSM_HDP_QUEUES=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_hdp_one_queue -o bench -- python3 $ROOT/bench.py --model hdp --no-cpu --steps 50 --warmup 10 --profile-steps 0 > $OUT/trace_hdp_one_queue.log 2>&1)
python3 profiles/iteration_timeline.py $OUT/trace_hdp_one_queue/bench_kernel_trace.csv > $OUT/hdp_timeline_one_queue.txt 2>&1
rm -rf $OUT/trace_hdp_one_queue
for m in lsm hdp cc ccu; do cp $OUT/kernel_stats_$m.csv profiles/${TAG}_kernel_stats_$m.csv; cp $OUT/traffic_$m.json profiles/${TAG}_traffic_$m.json; done
python3 profiles/kernel_durations.py $TAG > profiles/kernel_durations.json
python3 profiles/merge_traffic.py $TAG > profiles/traffic.json
cp profiles/kernel_durations.json profiles/traffic.json $OUT/
for m in lsm hdp cc ccu; do bash $ROOT/profiles/collect.sh $TAG $m bench >> $OUT/collect_$m.log 2>&1; done
cd $ROOT
python3 bench.py < /dev/null > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
: > $OUT/chains_per_gpu.jsonl
for C in 2 3 4; do python3 bench.py --chains-per-gpu $C --no-cpu >> $OUT/chains_per_gpu.jsonl 2>> $OUT/chains_per_gpu.err; done
python3 profiles/posterior_mixing.py > $OUT/posterior_mixing.txt 2>&1
# round 5: what a call of the device loops costs beyond its iterations; windows behind different predecessors;
# the case-control iteration launch by launch
python3 profiles/per_call_cost.py > $OUT/per_call_cost.jsonl 2> $OUT/per_call_cost.err
python3 profiles/window_probe.py > $OUT/window_probe.jsonl 2> $OUT/window_probe.err
python3 profiles/iteration_timeline.py $OUT/stats_cc/bench_kernel_trace.csv k_post_reduce_dir > $OUT/cc_timeline.txt 2>&1
python3 bench.py --model lsm --no-cpu --steps 20 --warmup 5 --windows 6 > $OUT/bench_windows_lsm.json 2> $OUT/bench_windows.err
python3 bench.py --model hdp --no-cpu --steps 20 --warmup 5 --windows 6 > $OUT/bench_windows_hdp.json 2>> $OUT/bench_windows.err
# round 6: what the case-control kernels execute (vector / scalar / LDS / memory instructions, busy and waiting
# clocks) - counters in runs of their own, as the traffic passes
: > $OUT/cc_pass_counters.jsonl
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_cc_pass -o p -- python3 $ROOT/bench.py --model cc --no-cpu --steps 4 --warmup 2 --profile-steps 0 --settle-steps 0 > $OUT/pmc_cc_pass.log 2>&1)
  python3 profiles/pass_counters.py $OUT/pmc_cc_pass >> $OUT/cc_pass_counters.jsonl 2>> $OUT/pmc_cc_pass.log
  rm -rf $OUT/pmc_cc_pass
done
python3 profiles/instr_counts.py > $OUT/instr_counts.json 2>&1
python3 profiles/instr_counts.py scratch > $OUT/hot_kernel_registers.txt 2>&1
python3 bench.py --gpus 2 --backend gloo --share-device0 --no-cpu 2> $OUT/bench_2ranks.err | grep "^{" > $OUT/bench_2ranks_one_gpu.json   # (gloo greets on stdout)
python3 bench.py --model lsm --cpu-procs 8 > $OUT/bench_lsm_cpu8.json 2> $OUT/bench_lsm_cpu8.err
if [ -f tmp_timing/libtiming.so ]; then
  python3 profiles/pipe_timing.py tmp_timing/libtiming.so $OUT/pipe_timing.json > $OUT/pipe_timing.log 2>&1
  python3 profiles/loglik_timing.py tmp_timing/libtiming.so $OUT/loglik_timing.json > $OUT/loglik_timing.log 2>&1
  python3 profiles/ccpipe_timing.py tmp_timing/libtiming.so $OUT/ccpipe_timing.json > $OUT/ccpipe_timing.log 2>&1
  python3 profiles/labels_phases.py tmp_timing/libtiming.so $OUT/labels_phases.json > $OUT/labels_phases.log 2>&1
  python3 profiles/hdp_tail_timing.py tmp_timing/libtiming.so $OUT/hdp_tail_timing.json > $OUT/hdp_tail_timing.log 2>&1
fi
[ -x tmp_timing/valu_rates ] && ./tmp_timing/valu_rates > $OUT/valu_rates.txt 2>&1
[ -x tmp_timing/sqrt_acc ] && ./tmp_timing/sqrt_acc > $OUT/sqrt_acc.txt 2>&1
python3 profiles/end_to_end_fit.py 1000 > $OUT/end_to_end_fit.jsonl 2> $OUT/end_to_end_fit.err
python3 profiles/end_to_end_fit.py 5000 >> $OUT/end_to_end_fit.jsonl 2>> $OUT/end_to_end_fit.err
ls -la $OUT | head -60
