"""Durations of the pipelined sweep's launches from a rocprofv3 kernel trace, grouped by
grid size (resolver-only launches have T workgroups, the others T + evaluators).

    python profiles/pipe_roles.py <kernel_trace.csv>
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if 'k_pipe_step' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
by_grid = defaultdict(list)
for r in rows:
    g = int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r['Grid_Size'])
    by_grid[g].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for g, v in sorted(by_grid.items()):
    v = sorted(v)
    print('grid %8d threads: n=%5d  median %.2f us  mean %.2f us' %
          (g, len(v), v[len(v) // 2] / 1e3, sum(v) / len(v) / 1e3))
gaps = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(rows, rows[1:])]
gaps = sorted(g for g in gaps if g < 50000)
if gaps:
    print('gap between consecutive k_pipe_step launches: median %.2f us' % (gaps[len(gaps) // 2] / 1e3))
