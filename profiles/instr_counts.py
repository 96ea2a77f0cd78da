"""Vector instructions per dyad term of the two hot loops, read from the BUILT library's code object
(bench.py's `f64_instr_per_term_in_kernel` / `frac_by_kernel_instructions` used hand-kept constants).

    python profiles/instr_counts.py [dynetlsm_amd/libdynetlsm_hip.so]

The gfx950 code object is cut out of the library's clang offload bundle and disassembled with
llvm-objdump; inside a kernel the neighbour trips are found by their v_rsq_f64 instructions
(one root per distance):
  * k_pipe_step<2, undirected, 1>: the fully unrolled no-flush loop - the run of equally long
    trips with two roots each (two positions of the moving node against 64 neighbours);
    per term = VALU instructions of a trip / 2;
  * k_loglik_undirected<2, 2>: the whole-tile loop - trips of four rows (four roots), two candidate
    intercepts sharing distance and exponential; per candidate-term = VALU of a trip / (4 * 2).
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


def extract_code_object(lib_path, arch='gfx950'):
    data = open(lib_path, 'rb').read()
    i = data.find(b'__CLANG_OFFLOAD_BUNDLE__')
    if i < 0:
        raise RuntimeError('no offload bundle in %s' % lib_path)
    n = struct.unpack_from('<Q', data, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, sz, tl = struct.unpack_from('<QQQ', data, off)
        off += 24
        triple = data[off:off + tl].decode()
        off += tl
        if arch in triple:
            return data[i + o:i + o + sz]
    raise RuntimeError('no %s code object in %s' % (arch, lib_path))


def disassemble(lib_path, with_addresses=False):
    """{mangled kernel name: [instruction text]} (with_addresses: [(byte address, text)])"""
    co = extract_code_object(lib_path)
    with tempfile.NamedTemporaryFile(suffix='.co', delete=False) as f:
        f.write(co)
        path = f.name
    try:
        txt = subprocess.run([OBJDUMP, '-d', '--mcpu=gfx950', '--no-show-raw-insn', path],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
    finally:
        os.unlink(path)
    funcs, name, cur = {}, None, []
    for line in txt.splitlines():
        m = re.match(r'^[0-9a-f]+ <([^>]+)>:$', line)
        if m:
            if name:
                funcs[name] = cur
            name, cur = m.group(1), []
        elif name and line.strip() and not line.startswith('Disassembly'):
            ins = line.strip().split('//')[0].strip()
            if ins:
                if with_addresses:
                    m2 = re.search(r'//\s*([0-9A-Fa-f]+):', line)
                    cur.append((int(m2.group(1), 16) if m2 else -1, ins))
                else:
                    cur.append(ins)
    if name:
        funcs[name] = cur
    return funcs


def _trips(ins, roots_per_trip):
    """(VALU count, length) of every stretch between two consecutive groups of roots"""
    pos = [k for k, l in enumerate(ins) if l.startswith('v_rsq_f64')]
    starts = pos[::roots_per_trip]
    out = []
    for a, b in zip(starts[:-1], starts[1:]):
        seg = ins[a:b]
        if sum(1 for l in seg if l.startswith('v_rsq_f64')) != roots_per_trip:
            continue
        out.append((sum(1 for l in seg if l.startswith('v_')), b - a,
                    sum(1 for l in seg if l.startswith(QUARTER_RATE))))
    return out


# issue cost of a vector instruction in cycles, measured on the MI355X with four wavefronts per SIMD
# (profiles/micro/valu_rates.hip -> profiles/r04_valu_rates.txt): v_fma_f64 and the other full-rate
# instructions 4.15, the quarter-rate float64 transcendentals 16.1
QUARTER_RATE = ('v_rsq_f64', 'v_rcp_f64', 'v_sqrt_f64')
CYCLES_FULL, CYCLES_QUARTER = 4.15, 16.1


def issue_slots(valu, quarter):
    """issue slots (units of one full-rate instruction) of `valu` vector instructions of which
    `quarter` are quarter-rate"""
    return (valu - quarter) + quarter * CYCLES_QUARTER / CYCLES_FULL


def _modal_run(trips, min_run):
    """the VALU count of the longest run of equally long trips (the unrolled steady state)"""
    best, i = None, 0
    while i < len(trips):
        j = i
        while j + 1 < len(trips) and abs(trips[j + 1][1] - trips[i][1]) <= 2:
            j += 1
        if j - i + 1 >= min_run and (best is None or trips[i][0] < best[0]):
            best = (trips[i][0], j - i + 1, trips[i][2])
        i = j + 1
    return best


def _rolled_trip_loops(ains):
    """(VALU, length, quarter-rate, SALU) of every loop (a backward branch and its target, by byte addresses: a
    branch's operand counts 4-byte words from the instruction behind it) that holds exactly two roots and stores
    nothing to memory: the bare trips of kernels_pipe_lds.hpp's evaluators (round 6: the neighbour loop is rolled -
    one trip of 64 neighbours at two positions per iteration)"""
    addr = [a for a, _ in ains]
    index = {a: i for i, a in enumerate(addr)}
    out = []
    for i, (a, l) in enumerate(ains):
        if not (l.startswith('s_cbranch') or l.startswith('s_branch')):
            continue
        t = l.split()[-1]
        if not t.isdigit() or int(t) < 32768:
            continue
        target = a + 4 + 4 * (int(t) - 65536)
        if target not in index:
            continue
        body = [x for _, x in ains[index[target]:i + 1]]
        nroot = sum(1 for x in body if x.startswith('v_rsq_f64'))
        if nroot not in (2, 4):                         # one trip per iteration, or two (the unrolled form)
            continue
        ntrip = nroot // 2
        if any(x.startswith('global_store') or x.startswith('s_barrier') for x in body):
            continue
        out.append((sum(1 for x in body if x.startswith('v_')) / ntrip, len(body) / ntrip,
                    sum(1 for x in body if x.startswith(QUARTER_RATE)) / ntrip,
                    sum(1 for x in body if x.startswith('s_') and not x.startswith(('s_waitcnt', 's_nop'))) / ntrip))
    return out


def counts(lib_path=None):
    lib_path = lib_path or os.path.join(ROOT, 'dynetlsm_amd', 'libdynetlsm_hip.so')
    funcs = disassemble(lib_path)
    sweep = next(v for k, v in funcs.items() if re.match(r'_ZN4dlsm11k_pipe_stepILi2ELi0ELi1E', k))
    ll = next(v for k, v in funcs.items() if re.match(r'_ZN4dlsm19k_loglik_undirectedILi2ELi2E', k))
    s = _modal_run(_trips(sweep, 2), 8)
    asweep = next(v for k, v in disassemble(lib_path, with_addresses=True).items()
                  if re.match(r'_ZN4dlsm11k_pipe_stepILi2ELi0ELi1E', k))
    rolled = _rolled_trip_loops(asweep)
    if rolled:          # the no-flush variant is the shortest of the three (flush counter / squared distances)
        r = min(rolled)
        s = (r[0], 1, r[2])
        salu_per_trip = r[3]
    else:
        salu_per_trip = None
    # the log-likelihood loops are rolled, one per variant (whole tile / ragged tile; the squared-
    # distance one has no root): a trip runs from the loop's head (behind the previous branch) to the
    # flush test behind its four roots (s_cmp_ge: the logarithms behind it run once per `nflush`
    # trips); the variant with the fewest vector instructions is the whole-tile one
    pos = [k for k, l in enumerate(ll) if l.startswith('v_rsq_f64')]
    bodies = []
    for g in range(0, len(pos) - 3, 4):
        a = pos[g]
        while a > 0 and not (ll[a - 1].startswith('s_cbranch') or ll[a - 1].startswith('s_branch')):
            a -= 1
        e = pos[g + 3]
        while e < len(ll) and not ll[e].startswith('s_cmp_ge'):
            e += 1
        bodies.append((sum(1 for l in ll[a:e] if l.startswith('v_')), e - a,
                       sum(1 for l in ll[a:e] if l.startswith(QUARTER_RATE))))
    body = min(bodies) if bodies else None
    out = {'library': os.path.relpath(lib_path, ROOT)}
    if s:
        out['k_pipe_step<2,0,1>'] = {'valu_per_trip_of_64_neighbours_2_positions': s[0],
                                    'loop': 'rolled (kernels_pipe_lds.hpp)' if rolled else 'unrolled',
                                    'salu_per_trip': salu_per_trip,
                                    'unrolled_trips_found': s[1], 'valu_per_term': round(s[0] / 2.0, 1),
                                    'quarter_rate_per_trip': s[2],
                                    'issue_slots_per_term': round(issue_slots(s[0], s[2]) / 2.0, 2)}
    try:
        cc = ccpipe_item_counts(lib_path)
        if cc:
            out['k_ccpipe_step<2>'] = cc
    except StopIteration:
        pass
    if body:
        out['k_loglik_undirected<2,2>'] = {'valu_per_trip_of_4_rows_2_candidates': body[0],
                                           'valu_per_candidate_term': round(body[0] / 8.0, 1),
                                           'quarter_rate_per_trip': body[2],
                                           'issue_slots_per_candidate_term':
                                               round(issue_slots(body[0], body[2]) / 8.0, 2)}
    return out


def ccpipe_item_counts(lib_path=None):
    """vector instructions of ONE evaluator item of the sparse case-control sweep (k_ccpipe_step<2>: a node of
    <= 256 gathered terms = one trip of four 64-term chunks + one flush of its window terms), from the code
    object: the chunk loop is peeled (first trip / later trips: eight roots each), the flush has two roots.
    An estimate with stated parts, for the bench line's issue-rate view of the kernel."""
    lib_path = lib_path or os.path.join(ROOT, 'dynetlsm_amd', 'libdynetlsm_hip.so')
    funcs = disassemble(lib_path)
    ins = next(v for k, v in funcs.items() if re.match(r'_ZN4dlsm13k_ccpipe_stepILi2E', k))
    pos = [k for k, l in enumerate(ins) if l.startswith('v_rsq_f64')]
    if len(pos) < 18:
        return None
    bar = [k for k, l in enumerate(ins[:pos[0]]) if l.startswith('s_barrier')]
    start = bar[-1] if bar else max(0, pos[0] - 200)         # behind the table's barrier: the item's prologue
    valu = lambda a, b: sum(1 for l in ins[a:b] if l.startswith('v_'))
    quarter = lambda a, b: sum(1 for l in ins[a:b] if l.startswith(QUARTER_RATE))
    first_trip = (valu(start, pos[8] - 20), quarter(start, pos[8] - 20))
    flush = (valu(pos[16] - 60, min(len(ins), pos[17] + 160)), quarter(pos[16] - 60, min(len(ins), pos[17] + 160)))
    v, q = first_trip[0] + flush[0], first_trip[1] + flush[1]
    return {'valu_per_item': v, 'quarter_rate_per_item': q, 'issue_slots_per_item': round(issue_slots(v, q), 1),
            'parts': {'prologue_and_first_trip_of_4_chunks': first_trip[0], 'flush_of_the_window_terms': flush[0]}}


def kernel_metadata(lib_path=None):
    """{demangled-ish kernel name: dict(vgpr, sgpr, vgpr_spill, sgpr_spill, scratch_bytes, lds, kernarg)}
    from the code object's notes (llvm-readelf --notes)"""
    lib_path = lib_path or os.path.join(ROOT, 'dynetlsm_amd', 'libdynetlsm_hip.so')
    co = extract_code_object(lib_path)
    with tempfile.NamedTemporaryFile(suffix='.co', delete=False) as f:
        f.write(co)
        path = f.name
    try:
        out = subprocess.run([os.path.join(os.path.dirname(OBJDUMP), 'llvm-readelf'), '--notes', path],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
        names = subprocess.run(['c++filt'],
                               input='\n'.join(re.findall(r'\.name:\s+(\S+)', out)).encode(),
                               stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
    finally:
        os.unlink(path)
    res = {}
    blocks = out.split('- .agpr_count')[1:]
    mangled = re.findall(r'\.name:\s+(\S+)', out)
    demangle = dict(zip(mangled, names))
    for b in blocks:
        m = re.search(r'\.name:\s+(\S+)', b)
        if not m:
            continue

        def g(k):
            q = re.search(r'\.%s:\s+(\d+)' % k, b)
            return int(q.group(1)) if q else 0
        nm = demangle.get(m.group(1), m.group(1))
        nm = nm.split('(')[0].replace('void ', '').replace('dlsm::', '').replace(' ', '')
        res[nm] = dict(vgpr=g('vgpr_count'), sgpr=g('sgpr_count'), vgpr_spill=g('vgpr_spill_count'),
                       sgpr_spill=g('sgpr_spill_count'), scratch_bytes=g('private_segment_fixed_size'),
                       lds=g('group_segment_fixed_size'), kernarg=g('kernarg_segment_size'))
    return res


# the kernels an iteration of the three benchmark configurations launches (D = 2): none of them
# may touch scratch memory (round-3 verdict: k_pipe_last_ride<2> had 12 scratch instructions)
HOT_KERNELS = [
    'k_pipe_step<2,0,1>', 'k_pipe_last_ride<2>', 'k_loglik_undirected<2,2>',
    'k_lsm_finalize_apply_propose<2>',
    'k_post_apply<2>', 'k_sample_labels_mfma<5>', 'k_label_counts', 'k_hdp_stage1<2>', 'k_hdp_stage2<2>',
    'k_hdp_stage3<2>', 'k_hdp_hypers_propose<2>', 'k_hdp_logp_batch_sums<2>', 'k_hdp_logp_batch_finish<2>',
    'k_ccpipe_step<2>', 'k_ccpipe_pack<2>', 'k_loglik_casecontrol_stream<2,1,false,2,1024>',
    'k_loglik_casecontrol_stream<2,4,false,1,1024>', 'k_cc_rows', 'k_cc_order',
    'k_post_reduce_dir<2>', 'k_post_apply_dir<2>', 'k_dir_reduce_accept_intercept<2>', 'k_dir_tail<2>',
]


def hot_scratch_report(lib_path=None):
    md = kernel_metadata(lib_path)
    return {k: md.get(k) for k in HOT_KERNELS}


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[-1] == 'scratch':         # python profiles/instr_counts.py scratch
        for k, v in hot_scratch_report().items():
            print('%-44s %s' % (k, v))
    else:
        print(json.dumps(counts(sys.argv[1] if len(sys.argv) > 1 else None), indent=1))
