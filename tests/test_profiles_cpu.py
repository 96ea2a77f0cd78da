"""The headline roofline of the committed bench line follows from the committed profiles
(round-3 verdict: `frac` divided by an event figure that no file under profiles/ held)."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, 'profiles')


def _latest(pattern):
    files = sorted(glob.glob(os.path.join(PROF, pattern)))
    return files[-1] if files else None


def test_headline_roofline_reproduces_from_the_stored_rocprof_average():
    bench = _latest('r??_bench_lsm.json')
    dur = os.path.join(PROF, 'kernel_durations.json')
    if bench is None or not os.path.exists(dur):
        pytest.skip('no committed bench line / kernel durations')
    line = json.loads(open(bench).read().strip().splitlines()[-1])
    rf = line['roofline']
    if 'us_per_launch_source' not in rf:
        pytest.skip('bench line older than the stored-duration convention')
    d = json.load(open(dur))
    tag = os.path.basename(bench)[:3]
    assert tag in d['_source'], (tag, d['_source'])          # durations and bench line of the same round
    if tag + '_kernel_stats' not in rf['us_per_launch_source']:
        pytest.skip('the committed bench line was computed from an earlier round\'s stored durations')
    avg_us = d['lsm']['k_pipe_step<2,0,1>']['avg_us']
    assert abs(rf['us_per_launch'] - avg_us) < 5e-3
    # T = 10, N = 2000: (proposal, current) x ordered pairs / 18 launches, 34 flop-slots x 2
    terms = 2.0 * 10 * 2000 * 1999 / rf['launches_per_sweep']
    assert abs(terms - rf['dyad_terms_per_launch']) < 1.0
    tflops = terms * rf['f64_ops_per_term'] * 2.0 / (avg_us * 1e-6) / 1e12
    assert abs(tflops / rf['peak'] - rf['frac']) < 2e-4, (tflops / rf['peak'], rf['frac'])
    assert abs(tflops - rf['achieved']) < 2e-2
    # the executed-instruction figure is the same duration priced by the code object's count
    ex = terms * rf['valu_instr_per_term_in_kernel'] * 2.0 / (avg_us * 1e-6) / 1e12 / rf['peak']
    assert abs(ex - rf['frac_executed']) < 2e-4
    assert rf['valu_instr_per_term_in_kernel'] < rf['f64_ops_per_term']       # what `peak_note` explains
    # the stored PMC traffic is the file's
    tr = json.load(open(os.path.join(PROF, 'traffic.json')))
    assert rf['traffic'] == tr['k_pipe_step']['hbm_bytes_per_launch']
