"""Spill hygiene of the built library (round-3 verdict item 9), read from its gfx950 code object:
no kernel an iteration of the three benchmark configurations launches may spill vector registers
or execute a scratch-memory instruction; the matrix-core label kernel is instantiated only for the
component counts that stay in registers (KS <= 8: K <= 32; above that capi.hip launches the
wavefront-per-node kernel)."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'profiles'))


@pytest.fixture(scope='module')
def code_object():
    import instr_counts as ic
    if not os.path.exists(ic.OBJDUMP):
        pytest.skip('llvm-objdump not at hand')
    from dynetlsm_amd.build import build
    lib = build()
    return ic, ic.kernel_metadata(lib), ic.disassemble(lib)


def _mangled(funcs, demangled_prefix):
    """entries of the disassembly whose demangled name starts with the prefix"""
    import subprocess
    names = list(funcs)
    dem = subprocess.run(['c++filt'], input='\n'.join(names).encode(), stdout=subprocess.PIPE,
                         check=True).stdout.decode().splitlines()
    out = []
    for m, d in zip(names, dem):
        d = d.split('(')[0].replace('void ', '').replace('dlsm::', '').replace(' ', '')
        if d == demangled_prefix:
            out.append(m)
    return out


def test_hot_kernels_use_no_scratch_memory(code_object):
    ic, md, funcs = code_object
    for name in ic.HOT_KERNELS:
        assert name in md, 'kernel %s is not in the library' % name
        assert md[name]['vgpr_spill'] == 0, (name, md[name])
        hits = _mangled(funcs, name)
        assert len(hits) == 1, (name, hits)
        n_scratch = sum(1 for line in funcs[hits[0]] if 'scratch_' in line)
        assert n_scratch == 0, '%s executes %d scratch instructions' % (name, n_scratch)
    # the single-chain sweep kernel and its batch form fit the 128 registers of a 1024-thread workgroup
    assert md['k_pipe_step<2,0,1>']['vgpr'] <= 128


def test_matrix_core_label_kernel_is_built_for_the_sizes_that_stay_in_registers(code_object):
    ic, md, funcs = code_object
    ks = sorted(int(re.match(r'k_sample_labels_mfma<(\d+)>', k).group(1)) for k in md
                if k.startswith('k_sample_labels_mfma<'))
    assert ks == list(range(1, 9)), ks
    for k in ks:
        m = md['k_sample_labels_mfma<%d>' % k]
        assert m['vgpr_spill'] == 0 and m['scratch_bytes'] == 0, (k, m)
