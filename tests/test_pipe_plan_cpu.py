"""The pipelined sweep's part plan (dynetlsm_amd/csrc/pipe_plan.hpp: which trips of 64 neighbours an evaluator item
takes; shared by the host, which fills the launch's plan table, and the device) checked without a GPU: the parts
partition a slice's trips for every network size, number of parts, batch and workgroup class, the window's trips are
the first trips of their parts, and nothing exceeds the LDS capacity the host sizes - under ASan / UBSan."""
import os

from test_sanitizers_cpu import SAN, _build_and_run


def test_pipe_plan_partitions_the_trips(tmp_path):
    out = _build_and_run(tmp_path, 'g++', [os.path.join(SAN, 'check_pipe_plan.cpp')], extra=['-std=c++17'])
    assert 'check_pipe_plan ok' in out
