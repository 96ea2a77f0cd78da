"""Build the gfx950 engine in-tree: dynetlsm_amd/libdynetlsm_hip.so.

hipcc cross-compiles without a GPU; the built library travels with the tree.
"""
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libdynetlsm_hip.so')
SOURCES = ['capi.hip']
INCLUDE = os.path.join(HERE, '..', 'include')
FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-Wno-unused-value',
         '-Wno-unused-result', '-shared', '-fPIC']


def hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', shutil.which('hipcc')):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC or install ROCm under /opt/rocm)')


def dependencies():
    """every file the library is compiled from: all of csrc/ and include/ (capi.hip is a
    unity build that includes every header, so any of them can change the binary)"""
    deps = []
    for pat in (os.path.join(CSRC, '*.hip'), os.path.join(CSRC, '*.hpp'),
                os.path.join(CSRC, '*.h'), os.path.join(INCLUDE, '*.h')):
        deps.extend(glob.glob(pat))
    return sorted(deps)


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in dependencies())


def build(force=False, verbose=False):
    """Compile every HIP source of the engine for gfx950."""
    if not force and not stale():
        return LIB
    cmd = [hipcc()] + FLAGS + ['-o', LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
