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


def _header_arity():
    """{name: number of parameters} of every entry point declared in the header"""
    src = open(os.path.join(ROOT, 'include', 'dynetlsm_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for name, args in re.findall(r'\b(dlsm_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        args = args.strip()
        out[name] = 0 if args in ('', 'void') else len(args.split(','))
    return out


def _split_args(s):
    """top-level comma split of a call's argument text"""
    parts, depth, cur = [], 0, ''
    for ch in s:
        if ch in '([{':
            depth += 1
        elif ch in ')]}':
            depth -= 1
        if ch == ',' and depth == 0:
            parts.append(cur)
            cur = ''
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return parts


def test_integration_document_quotes_the_header_signatures():
    """every `dlsm_*(...)` call INTEGRATION.md shows a maintainer must name an entry point of the
    header with the header's number of arguments, and every `lib.dlsm_*.argtypes = [...]` row must
    have that many entries (round-3 verdict: two quoted signatures had drifted from the header)"""
    arity = _header_arity()
    assert len(arity) >= 70
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    calls = []
    for m in re.finditer(r'\b(dlsm_[a-z_0-9]+)\(', doc):
        name, i, depth = m.group(1), m.end(), 1
        j = i
        while j < len(doc) and depth:
            depth += doc[j] in '([{'
            depth -= doc[j] in ')]}'
            j += 1
        calls.append((name, doc[i:j - 1]))
    assert len(calls) >= 25
    for name, args in calls:
        assert name in arity, 'INTEGRATION.md names %s, which the header does not declare' % name
        n = len(_split_args(args))
        if args.strip() in ('', '...'):
            continue
        assert n == arity[name], ('INTEGRATION.md quotes %s(%s): %d arguments, the header has %d'
                                  % (name, ' '.join(args.split()), n, arity[name]))
    # every name the document mentions at all exists
    for name in set(re.findall(r'\b(dlsm_[a-z_0-9]+)\b', doc)):
        if name.endswith('_'):
            continue
        assert name in arity or name in ('dlsm_hdp_config', 'dlsm_lsm_config', 'dlsm_chain'), name
    # the ctypes rows: count the entries of the list literal
    for m in re.finditer(r'lib\.(dlsm_[a-z_0-9]+)\.argtypes\s*=\s*(.+)', doc):
        name, rhs = m.group(1), m.group(2)
        if '\n' in rhs:
            rhs = rhs.split('\n')[0]
        # rows that continue on the next line end in a comma inside the bracket
        k = m.end()
        while rhs.count('[') > rhs.count(']'):
            nl = doc.index('\n', k + 1) if '\n' in doc[k + 1:] else len(doc)
            rhs += doc[k:nl]
            k = nl
        rhs = re.sub(r'#.*', '', rhs)
        n = 0
        for piece, mult in re.findall(r'\[([^\]]*)\](?:\s*\*\s*(\d+))?', rhs):
            cnt = len([a for a in _split_args(piece) if a.strip()])
            n += cnt * (int(mult) if mult else 1)
        assert n == arity[name], ('INTEGRATION.md binds %s with %d argtypes, the header has %d'
                                  % (name, n, arity[name]))


def test_n_features_beyond_the_kernels_is_a_value_error_before_any_device_call():
    """the reference takes any n_features (lsm.py:235,254); the kernels are instantiated for 1..8:
    fit() says so by name, on a box without a GPU too (no device call has been made yet)"""
    import numpy as np
    import dynetlsm_amd as da
    Y = np.zeros((2, 6, 6))
    for est in (da.DynamicNetworkLSM(n_features=9, n_iter=3, tune=None, burn=None),
                da.DynamicNetworkHDPLPCM(n_features=12, n_iter=3, tune=None, burn=None),
                da.DynamicNetworkLPCM(n_features=0, n_iter=3, tune=None, burn=None)):
        with pytest.raises(ValueError, match='1 <= n_features <= 8'):
            est.fit(Y)
    with pytest.raises(ValueError, match='n_features=9'):
        da.Chain(2, 6, 9, 'undirected')
