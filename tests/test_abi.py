"""CPU-side checks of the drop-in boundary: the shared library loads and exports
every symbol declared in include/rvsgpu.h, argument validation works without a
GPU, and the product path refuses to run without one (no CPU fallback)."""
import os
import re
import subprocess

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    so = os.path.join(REPO, 'rvspecfit_amd', 'librvsgpu.so')
    if not os.path.exists(so):
        subprocess.check_call(['make', '-C',
                               os.path.join(REPO, 'rvspecfit_amd', 'csrc'), '-j8'],
                              stdout=subprocess.DEVNULL)
    from rvspecfit_amd import _lib
    return _lib.lib()


def header_symbols():
    txt = open(os.path.join(REPO, 'include', 'rvsgpu.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(rvs_[a-z0-9_]+)\s*\(', txt)))


def test_every_declared_symbol_is_exported(lib):
    from rvspecfit_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), s
        assert s in _lib.SIGNATURES, 'ctypes signature missing for ' + s
    for s in _lib.SIGNATURES:
        assert s in syms, 'binding without a header declaration: ' + s


def test_abi_version_and_work_size(lib):
    assert lib.rvs_abi_version() == 1
    assert lib.rvs_chisq_work_size(100, 3) == 100 + 2 * 3 * 100 + 2 * 3


def test_argument_validation_without_gpu(lib):
    # shape errors are detected on the host before any launch
    assert lib.rvs_spline_construct(None, None, 2, 1, 0, None, None, None) == -1
    assert lib.rvs_spline_construct(None, None, 9, 1, 3, None, None, None) == -1
    assert lib.rvs_chisq_grid(None, None, None, 10, 99, 1, None, None, 10, 1, 1,
                              None, None, 1, None, 0, 4, None, 0., 0., 0, None,
                              None, None) == -1
    assert lib.rvs_ccf_xcorr(None, None, 1000, 1, None, None, 1, None, 1, None,
                             None, 5, None, None, 5, 0., None, None, None,
                             None) == -1   # nfft not a power of two
    k3 = np.array([1.0, 2.0, 4.1])
    assert lib.rvs_chisq_prepare(None, None, None, 10, 1, k3.ctypes.data, 1, 0.,
                                 k3.ctypes.data, None) == -3  # evaler's -2


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from rvspecfit_amd import _lib, spec_fit
    with pytest.raises(_lib.RvsGpuError):
        spec_fit.convolve_vsini(np.exp(np.linspace(1, 1.1, 50)), np.ones(50), 10.)


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, 'rvspecfit_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                txt = open(os.path.join(root, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, f
                assert 'oracle_core' not in txt, f


def test_struct_layouts_match_the_header(tmp_path):
    """the ctypes mirrors of the header's structs (rvspecfit_amd/_lib.py) have
    the size and field offsets a C compiler gives them -- which also shows that
    include/rvsgpu.h is plain C"""
    import ctypes
    from rvspecfit_amd import _lib
    pairs = [('rvs_point_arm', _lib.PointArm),
             ('rvs_objective_arm', _lib.ObjectiveArm),
             ('rvs_nm_state', _lib.NmState),
             ('rvs_nm_objective', _lib.NmObjective)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rvsgpu.h"',
             'int main(void) {']
    for cname, cls in pairs:
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));'
                         % (cname, f[0], cname, f[0]))
    lines += ['return 0; }']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = str(tmp_path / 'layout')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-I',
                           os.path.join(REPO, 'include'), str(src), '-o', exe])
    got = dict(l.split() for l in
               subprocess.check_output([exe]).decode().splitlines())
    for cname, cls in pairs:
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for f in cls._fields_:
            assert int(got['%s.%s' % (cname, f[0])]) == \
                getattr(cls, f[0]).offset, (cname, f[0])
