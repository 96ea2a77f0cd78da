"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; each with
--kernel-trace, CSV output) into per-launch HBM bytes per kernel, corrected as
MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE / WRITE_SIZE are in
KiB-like units of the memory-side request counters (x1024 bytes) and on gfx950
FETCH_SIZE reads exactly 1/2 of a wide coalesced read stream -> doubled (an
upper bound for narrower accesses, which are uncalibrated).

    python profiles/pmc_traffic.py <fetch_dir> <write_dir> > profiles/traffic.json
"""
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') != counter:
                continue
            name = row['Kernel_Name'].split('(')[0].split('<')[0].split('::')[-1].strip()
            name = name.replace('void ', '')
            s = out.setdefault(name, [0.0, 0])
            s[0] += float(row['Counter_Value'])
            s[1] += 1
    return out


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, [0.0, 0])
        w, nw = write.get(k, [0.0, 0])
        fb = 2.0 * 1024.0 * f / max(nf, 1)      # gfx950 1/2 correction
        wb = 1024.0 * w / max(nw, 1)
        res[k] = {'launches_profiled': max(nf, nw),
                  'fetch_bytes_per_launch_corrected': round(fb, 1),
                  'write_bytes_per_launch': round(wb, 1),
                  'hbm_bytes_per_launch': round(fb + wb, 1)}
    json.dump(res, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
