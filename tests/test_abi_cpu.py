"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/dynetlsm_hip.h declares; no compute without a GPU;
the product never routes through the oracle."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'dynetlsm_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dlsm_[a-z_0-9]+)\s*\(', src)))


def test_library_builds_and_exports_every_declared_symbol():
    from dynetlsm_amd.build import build
    lib = ctypes.CDLL(build())
    names = _declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), 'missing export %s' % n
    assert lib.dlsm_abi_version() == 1


def test_stale_tracks_every_source_of_the_library():
    """build.stale() must notice a change to ANY file the unity build includes
    (round-1 advice: the headline kernel's header was not watched)"""
    from dynetlsm_amd import build as b
    b.build()
    deps = b.dependencies()
    names = {os.path.basename(d) for d in deps}
    for must in ('capi.hip', 'kernels_spec_pipe.hpp', 'kernels_hdp.hpp', 'kernels_dirloop.hpp',
                 'host_draws.hpp', 'capi_init.hpp', 'dynetlsm_hip.h', 'device_common.hpp'):
        assert must in names, must
    # every header the sources include is a dependency
    inc = set()
    for d in deps:
        inc.update(re.findall(r'#include\s+"([^"]+)"', open(d).read()))
    assert {os.path.basename(i) for i in inc} <= names
    assert not b.stale()
    target = [d for d in deps if d.endswith('kernels_spec_pipe.hpp')][0]
    st = os.stat(target)
    try:
        os.utime(target, (st.st_atime, os.path.getmtime(b.LIB) + 10))
        assert b.stale()
    finally:
        os.utime(target, (st.st_atime, st.st_mtime))
    assert not b.stale()


def test_binding_covers_the_header():
    from dynetlsm_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    _lib.load()


def test_no_device_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from dynetlsm_amd import Chain, EngineError
    with pytest.raises(EngineError) as e:
        Chain(2, 10, 2, 'undirected')
    assert e.value.code == -3


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'dynetlsm_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in txt.replace('the oracle', '').replace(
                    'CPU oracle', '').replace('oracle (', ''), os.path.join(dirpath, f)


def test_imputer_matches_reference():
    """imputer.py:11-81 (SimpleNetworkImputer, strategy='random'): bit for bit"""
    import numpy as np
    from conftest import load_golden
    from dynetlsm_amd.imputer import SimpleNetworkImputer
    g = load_golden('imputer.npz')
    for tag in 'ud':
        got = SimpleNetworkImputer(strategy='random', missing_value=-1).fit_transform(g[tag + '_Y'])
        np.testing.assert_array_equal(got, g[tag + '_random'])
