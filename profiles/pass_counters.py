"""Per-launch averages of the counters rocprofv3 collected (--pmc ... --output-format csv) for the kernels whose names
contain one of the given substrings.

    python profiles/pass_counters.py <rocprofv3 output dir> [name substring ...]

Used by collect_round.sh for the case-control model's kernels (profiles/r06_cc_pass_notes.md: SQ_INSTS_VALU is what showed
round 5's likelihood pass bound by vector issue).  Counter values are sums over the device; SQ_WAVE_CYCLES,
SQ_ACTIVE_INST_VALU and SQ_WAIT_INST_ANY count in units of four clocks.
"""
import collections
import csv
import glob
import json
import sys


def main():
    d = sys.argv[1]
    pats = sys.argv[2:] or ['loglik_casecontrol', 'ccpipe_step']
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if any(p in k for p in pats):
                acc[k.split('(')[0].replace('void dlsm::', '')][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in sorted(acc.items()):
        print(json.dumps({'kernel': k, 'launches': len(next(iter(cs.values()))),
                          **{c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}}))


if __name__ == '__main__':
    main()
