"""Timeline of one iteration of a device-resident loop from a rocprofv3 kernel trace: every launch
between two consecutive launches of an anchor kernel (default: the sweep's last launch), with its
queue, its start and end relative to the anchor and the gap to the previous launch on its queue -
medians over the iterations of the trace.  Made for the HDP-LPCM loop's two queues (the intercept's
likelihood pass beside the label update and the conjugate draws).
    python profiles/iteration_timeline.py <kernel_trace.csv> [anchor substring]
"""
import csv
import re
import sys

import numpy as np

anchor = sys.argv[2] if len(sys.argv) > 2 else 'k_pipe_last_ride'
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'dlsm::', '', name)
    return name.split('(')[0]


idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
if len(idx) < 4:
    sys.exit('anchor %r: fewer than 4 launches in the trace' % anchor)
# iterations = [anchor_i, anchor_{i+1}); keep those with the modal launch count
its = [rows[a:b] for a, b in zip(idx[:-1], idx[1:])]
counts = np.array([len(x) for x in its])
mode = np.bincount(counts).argmax()
its = [x for x in its if len(x) == mode][2:]
sig = [short(r['Kernel_Name']) for r in its[0]]
its = [x for x in its if [short(r['Kernel_Name']) for r in x] == sig]
start = np.array([[int(r['Start_Timestamp']) - int(x[0]['Start_Timestamp']) for r in x] for x in its]) / 1e3
end = np.array([[int(r['End_Timestamp']) - int(x[0]['Start_Timestamp']) for r in x] for x in its]) / 1e3
period = np.array([int(b[0]['Start_Timestamp']) - int(a[0]['Start_Timestamp']) for a, b in zip(its[:-1], its[1:])
                   if int(b[0]['Dispatch_Id']) - int(a[0]['Dispatch_Id']) == mode]) / 1e3
queues = [r['Queue_Id'] for r in its[0]]
print('%d iterations of %d launches; anchor %s; period median %.1f us' % (len(its), mode, anchor,
                                                                        np.median(period) if period.size else -1))
print('%-44s %5s %9s %9s %8s %8s' % ('kernel', 'queue', 'start', 'end', 'dur', 'gap'))
last_end = {}
for j, name in enumerate(sig):
    s, e = np.median(start[:, j]), np.median(end[:, j])
    q = queues[j]
    gap = np.median(start[:, j] - last_end[q]) if q in last_end else float('nan')
    last_end[q] = end[:, j]
    print('%-44s %5s %9.1f %9.1f %8.1f %8.1f' % (name[:44], q, s, e, np.median(end[:, j] - start[:, j]), gap))
