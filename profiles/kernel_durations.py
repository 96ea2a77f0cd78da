"""rocprofv3 --kernel-trace --stats summaries (profiles/rNN_kernel_stats_{lsm,hdp,cc}.csv) ->
profiles/kernel_durations.json: calls and average duration per kernel instantiation, the figure
`bench.py` divides the algorithmic flop by for `roofline.frac` (so that the fraction in the bench
line follows from the files under profiles/; the duration the run itself measures with events on
the launches is reported beside it).

    python profiles/kernel_durations.py r04 > profiles/kernel_durations.json
"""
import csv
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def short(name):
    """'void dlsm::k_pipe_step<2, 0, 1>(dlsm::ChainView, ...)' -> 'k_pipe_step<2,0,1>'"""
    n = name.split('(')[0].replace('void ', '').strip()
    n = n.split('::')[-1] if '<' not in n else re.sub(r'^.*?::(?=[A-Za-z_0-9]+<)', '', n)
    return n.replace(' ', '')


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
    out = {'_source': 'profiles/%s_kernel_stats_{lsm,hdp,cc}.csv (rocprofv3 --kernel-trace --stats of '
                      'bench.py --model M --no-cpu --steps 50 --warmup 10 --profile-steps 0)' % tag}
    for model in ('lsm', 'hdp', 'cc', 'ccu'):
        path = os.path.join(HERE, '%s_kernel_stats_%s.csv' % (tag, model))
        if not os.path.exists(path):
            continue
        rows = {}
        for r in csv.DictReader(open(path)):
            rows[short(r['Name'])] = {'calls': int(r['Calls']), 'avg_us': round(float(r['AverageNs']) / 1e3, 4),
                                      'pct': float(r['Percentage'])}
        out[model] = rows
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
