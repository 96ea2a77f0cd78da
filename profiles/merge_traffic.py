"""profiles/<tag>_traffic_{lsm,hdp,cc}.json (pmc_traffic.py, one per workload) -> the file bench.py reads: per kernel
the entry with the most profiled launches.
    python profiles/merge_traffic.py r04 > profiles/traffic.json
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
m = {}
for model in ('lsm', 'hdp', 'cc', 'ccu'):
    path = os.path.join(HERE, '%s_traffic_%s.json' % (tag, model))
    if not os.path.exists(path):
        continue
    for k, v in json.load(open(path)).items():
        if k not in m or v['launches_profiled'] > m[k]['launches_profiled']:
            m[k] = v
m['_source'] = ('profiles/%s_pmc_{fetch,write}_size_{lsm,hdp,cc}.csv through profiles/pmc_traffic.py '
                '(FETCH_SIZE x2 on gfx950)' % tag)
json.dump(m, sys.stdout, indent=1, sort_keys=True)
