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


# Every kernel of the library outside the list above: scratch memory is tolerated only where it is known, off
# the benchmark configurations' paths, and may not grow (round-4 verdict, weak 10; round 5 removed the
# two-batch and persistent sweeps and cleared every k_ccpipe_step / k_loglik_casecontrol instantiation).
# bytes of private segment per thread at HEAD:
KNOWN_SCRATCH = {
    # d = 3 / 4 instantiations of the pipelined sweep (resolver: the blocks' 64 registers + 8 d of the owners)
    'k_pipe_step<3,1,1>': 12, 'k_pipe_step<3,2,1>': 52, 'k_pipe_step<4,0,1>': 20, 'k_pipe_step<4,3,1>': 52,
    'k_pipe_step<4,1,1>': 60, 'k_pipe_step<4,2,1>': 84, 'k_pipe_last_ride<4>': 36,
    # d = 2: the dense case-control form (512 <= N < 2048).  (Round 6: the fallback for parts longer than the prefetch,
    # k_pipe_step<2,3,1>, is clean - its rare-path exponential is the table's, as the LDS evaluators'.)
    'k_pipe_step<2,2,1>': 28,
    # initialisation pipeline at d = 3 / 4 (one workgroup, once per fit: d x d Jacobi on indexed local arrays)
    'k_gmds_finish<3>': 548, 'k_gmds_finish<4>': 1448, 'k_lanczos_init<4>': 36, 'k_lanczos_step<4>': 32,
    'k_partial_all<3>': 32,
    # 8 bytes of an indexed local array each, no spilled register
    'k_hdp_hypers': 8,
    **{'k_hdp_hypers_propose<%d>' % d: 8 for d in range(1, 9)},
    **{'k_hdp_logp_batch_finish<%d>' % d: 8 for d in range(1, 9)},
    # n_features 5 .. 8 (built in round 5; off every benchmark configuration).  The initialisation pipeline, once
    # per fit: Lanczos vectors and the d x d Jacobi of one 1024-thread workgroup
    'k_lanczos_init<5>': 164, 'k_lanczos_step<5>': 160, 'k_gmds_finish<5>': 2596,
    'k_lanczos_init<6>': 292, 'k_lanczos_step<6>': 288, 'k_gmds_finish<6>': 4000,
    'k_lanczos_init<7>': 420, 'k_lanczos_step<7>': 416, 'k_gmds_finish<7>': 5672,
    'k_lanczos_init<8>': 548, 'k_lanczos_step<8>': 544, 'k_gmds_finish<8>': 7592,
    # the per-node partials of the function seam (an indexed local array, no spilled register) and the slice
    # sweep's two registers beyond the 128 of its 1024-thread workgroup at d = 7, 8
    'k_partial_all<5>': 48, 'k_partial_all<6>': 64, 'k_partial_all<7>': 64,
    'k_sweep_slice<7,0>': 12, 'k_sweep_slice<8,0>': 12,
    # the pipelined sweep at d = 5 .. 8 (its register plan is d = 2's: the item's own 4 d registers of positions and
    # two or three prefetched trips of d registers each no longer fit the 128 of a 1024-thread workgroup; 2910 it/s at
    # d = 5, 1465 at d = 8 against 1660 / 1330 for the speculative sweep - DESIGN.md 9)
    'k_pipe_last_ride<5>': 76, 'k_pipe_last_ride<6>': 148, 'k_pipe_last_ride<7>': 300, 'k_pipe_last_ride<8>': 500,
    'k_pipe_step<5,0,1>': 72, 'k_pipe_step<5,1,1>': 108, 'k_pipe_step<5,2,1>': 108, 'k_pipe_step<5,3,1>': 84,
    'k_pipe_step<6,0,1>': 116, 'k_pipe_step<6,1,1>': 180, 'k_pipe_step<6,2,1>': 164, 'k_pipe_step<6,3,1>': 144,
    'k_pipe_step<7,0,1>': 172, 'k_pipe_step<7,1,1>': 224, 'k_pipe_step<7,2,1>': 208, 'k_pipe_step<7,3,1>': 200,
    'k_pipe_step<8,0,1>': 212, 'k_pipe_step<8,1,1>': 312, 'k_pipe_step<8,2,1>': 248, 'k_pipe_step<8,3,1>': 240,
    # ... and the sparse case-control sweep there (two chunks of records and proposals of d doubles in flight)
    'k_ccpipe_step<5>': 24, 'k_ccpipe_step<6>': 72, 'k_ccpipe_step<7>': 136, 'k_ccpipe_step<8>': 168,
}
# vector registers parked in accumulation registers (no memory traffic: scratch_bytes is 0)
KNOWN_AGPR_PARKED = {'k_post_apply<8>', 'k_lsm_finalize_apply_propose<8>', 'k_post_apply_dir<8>', 'k_post_align<8>'}


def test_no_kernel_outside_the_known_list_touches_scratch_memory(code_object):
    ic, md, funcs = code_object
    over = {k: v['scratch_bytes'] for k, v in md.items()
            if v['scratch_bytes'] > KNOWN_SCRATCH.get(k, 0) or
            (v['vgpr_spill'] and k not in KNOWN_SCRATCH and k not in KNOWN_AGPR_PARKED)}
    assert not over, 'kernels with more scratch memory than recorded: %s' % over
    # the list does not rot: an entry whose kernel is clean now must be removed
    stale = [k for k in KNOWN_SCRATCH if k in md and md[k]['scratch_bytes'] == 0]
    assert not stale, 'clean now, remove from KNOWN_SCRATCH: %s' % stale
    # what a sweep can launch at d = 2 with the defaults (resolve_sweep_algo: 1, 2, 4, 5) and the case-control passes
    for k in ('k_sweep_slice<2,0>', 'k_sweep_slice<2,1>', 'k_sweep_casecontrol<2>', 'k_pipe_step<2,0,1>',
              'k_pipe_step<2,1,1>', 'k_ccpipe_step<1>', 'k_ccpipe_step<2>', 'k_ccpipe_step<3>', 'k_ccpipe_step<4>',
              'k_loglik_casecontrol_rows<2,1>', 'k_loglik_casecontrol_rows<2,2>', 'k_cc_rows', 'k_cc_order',
              'k_loglik_casecontrol_stream<2,1,false,2,1024>', 'k_loglik_casecontrol_stream<2,4,false,1,1024>',
              'k_loglik_casecontrol_stream<2,2,false,1,1024>', 'k_loglik_casecontrol_stream<2,2,true,1,256>',
              'k_loglik_casecontrol_stream<2,1,false,1,256>', 'k_loglik_casecontrol_stream<2,4,false,1,256>'):
        assert k in md, k
        assert md[k]['scratch_bytes'] == 0 and md[k]['vgpr_spill'] == 0, (k, md[k])
