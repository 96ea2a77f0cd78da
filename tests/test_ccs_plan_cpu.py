"""The case-control likelihood pass's plan arithmetic (dynetlsm_amd/csrc/ccs_plan.hpp: the walking order's key and
entries - cc_rows.hpp, k_cc_order -, the wavefronts' shares of a slice's entries and the workgroups per slice of
k_loglik_casecontrol_stream; shared by the host and the device) checked without a GPU, under ASan / UBSan: the key fits
32 bits inside its bounds and orders rows by (out-degree, controls), a row's entries cover its out-terms, the shares
partition a slice's entries also where the product needs 64 bits, the trimmed grid stays resident and inside the
records' room, and the order built by counting lists every row's segments once, rows by descending key."""
import os

from test_sanitizers_cpu import SAN, _build_and_run


def test_ccs_plan_partitions_and_orders(tmp_path):
    out = _build_and_run(tmp_path, 'g++', [os.path.join(SAN, 'check_ccs_plan.cpp')], extra=['-std=c++17'])
    assert 'check_ccs_plan ok' in out
