"""Copy the judged artefacts of a `bash profiles/collect_round.sh <tag>` run (merged back into
gpurun_out/<tag>/) into profiles/ as <tag>_*: bench lines, rocprofv3 stats, PMC passes, stored durations
and traffic, chains-per-GPU lines, timing jsons of the -DDLSM_PIPE_TIMING build when it was at hand.
    python profiles/import_round.py r04
"""
import glob
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
tag = sys.argv[1]
src = os.path.join(os.path.dirname(HERE), 'gpurun_out', tag)
n = 0
for m in ('lsm', 'hdp', 'cc', 'ccu'):
    for a in ('bench_%s.json', 'kernel_stats_%s.csv', 'traffic_%s.json', 'pipe_roles_%s.txt'):
        p = os.path.join(src, a % m)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copy(p, os.path.join(HERE, '%s_%s' % (tag, a % m))); n += 1
    for kind in ('fetch', 'write'):
        f = glob.glob(os.path.join(src, 'pmc_%s_%s' % (kind, m), '*counter_collection.csv'))
        if f:
            shutil.copy(f[0], os.path.join(HERE, '%s_pmc_%s_size_%s.csv' % (tag, kind, m))); n += 1
for a in ('bench_default.json', 'bench_driver_args.json', 'chains_per_gpu.jsonl',
          'bench_2ranks_one_gpu.json', 'bench_lsm_cpu8.json', 'posterior_mixing.txt', 'instr_counts.json',
          'hot_kernel_registers.txt', 'end_to_end_fit.jsonl', 'pipe_timing.json', 'loglik_timing.json',
          'ccpipe_timing.json', 'labels_phases.json', 'hdp_tail_timing.json',
          'valu_rates.txt', 'sqrt_acc.txt', 'hdp_timeline_two_queues.txt', 'hdp_timeline_one_queue.txt',
          'per_call_cost.jsonl', 'window_probe.jsonl', 'cc_timeline.txt', 'bench_windows_lsm.json',
          'bench_windows_hdp.json', 'cc_pass_counters.jsonl'):
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(HERE, '%s_%s' % (tag, a))); n += 1
for a in ('kernel_durations.json', 'traffic.json'):
    p = os.path.join(src, a)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(HERE, a)); n += 1
print('%d files imported from %s' % (n, src))
