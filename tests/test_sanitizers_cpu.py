"""CPU-only sanitizer job (round-1 verdict, build hygiene): the scalar C oracle and the
product's GPU-free host C++ (native table draws, tridiagonal eigen-solver) compiled with
AddressSanitizer + UndefinedBehaviorSanitizer and run on small inputs.  (GPU sanitizers are
not available on the pool: the CPU build is where they run.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, 'oracle', 'sanitize')
FLAGS = ['-g', '-O1', '-fno-omit-frame-pointer', '-fsanitize=address,undefined',
         '-fno-sanitize-recover=undefined']
ENV = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0',
           UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')


def _build_and_run(tmp_path, compiler, sources, extra=()):
    if shutil.which(compiler) is None:
        pytest.skip('%s not installed' % compiler)
    exe = str(tmp_path / 'san')
    cmd = [compiler] + FLAGS + list(extra) + sources + ['-o', exe, '-lm']
    b = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert b.returncode == 0, b.stdout.decode()
    r = subprocess.run([exe], env=ENV, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=300)
    out = r.stdout.decode()
    assert r.returncode == 0, out
    assert 'runtime error' not in out and 'AddressSanitizer' not in out, out
    return out


def test_oracle_c_under_asan_ubsan(tmp_path):
    out = _build_and_run(tmp_path, 'gcc', [os.path.join(SAN, 'sanitize_oracle.c'),
                                           os.path.join(ROOT, 'oracle', 'dynetlsm_oracle.c')],
                         extra=['-std=gnu11'])
    assert 'sanitize_oracle ok' in out


def test_host_cpp_under_asan_ubsan(tmp_path):
    out = _build_and_run(tmp_path, 'g++', [os.path.join(SAN, 'sanitize_host.cpp')],
                         extra=['-std=c++17'])
    assert 'sanitize_host ok' in out
