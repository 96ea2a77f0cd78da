"""Durations of consecutive launches of one kernel from a rocprofv3 kernel trace, folded by
position inside a sweep (the pipelined sweeps launch a fixed sequence per iteration: the
first launch only evaluates, the last only resolves).
    python profiles/launch_sequence.py <kernel_trace.csv> <kernel substring> <launches per sweep>
"""
import csv
import sys

import numpy as np

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
per = int(sys.argv[3])
d = np.array([int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]) / 1e3
n = d.size // per * per
d = d[:n].reshape(-1, per)
grid = [r.get('Grid_Size', r.get('Grid_Size_X', '?')) for r in rows[:per]]
for i in range(per):
    print('launch %2d  grid %8s  median %7.2f us  mean %7.2f us' % (i, grid[i], np.median(d[:, i]),
                                                                   d[:, i].mean()))
print('sum of medians %.1f us over %d sweeps' % (np.median(d, axis=0).sum(), d.shape[0]))
