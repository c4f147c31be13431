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
    from rvspecfit_amd import _lib
    hdr = open(os.path.join(REPO, 'include', 'rvsgpu.h')).read()
    ver = int(re.search(r'#define RVS_ABI_VERSION (\d+)', hdr).group(1))
    assert lib.rvs_abi_version() == ver == _lib.ABI_VERSION
    assert lib.rvs_chisq_work_size(100, 3) == 100 + 4 * 3 * 100 + 2 * 3 + 2 * 100


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


def test_option_table(lib):
    """rvs_option_set / _get: a host-side table (no GPU needed), unknown names are
    argument errors, `with _lib.option(...)` restores the previous value; the
    table is filled from the environment ONCE -- a later setenv changes nothing"""
    import ctypes
    from rvspecfit_amd import _lib
    v = ctypes.c_int(-1)
    assert lib.rvs_option_get(b'obj_inblk_max', ctypes.byref(v)) == 0
    assert v.value == int(os.environ.get('RVS_OBJ_INBLK_MAX', 768))
    for name in ('xc_ws', 'xc_ws1', 'nm_glue', 'nm_bucket', 'obj_sort', 'nn_pipe',
                 'nm_split_min', 'nm_spec_max', 'nm_tail_window'):
        assert lib.rvs_option_get(name.encode(), ctypes.byref(v)) == 0
    assert lib.rvs_option_get(b'no_such_switch', ctypes.byref(v)) == -1
    assert lib.rvs_option_set(b'no_such_switch', 1) == -1
    assert lib.rvs_option_get(b'xc_ws', None) == -1
    before = _lib.set_option('obj_sort', 1)
    with _lib.option('obj_sort', 0):
        lib.rvs_option_get(b'obj_sort', ctypes.byref(v))
        assert v.value == 0
        os.environ['RVS_OBJ_SORT'] = '1'      # read once: no effect any more
        lib.rvs_option_get(b'obj_sort', ctypes.byref(v))
        assert v.value == 0
        del os.environ['RVS_OBJ_SORT']
    lib.rvs_option_get(b'obj_sort', ctypes.byref(v))
    assert v.value == 1
    _lib.set_option('obj_sort', before)


def test_basis_build_limits(lib):
    """rvs_basis_build takes what rvs_chisq_full takes (npoly <= 32) and arms of up
    to 16384 pixels; beyond that an argument error, before any launch"""
    one = np.ones(4)
    p = one.ctypes.data
    assert lib.rvs_basis_build(p, None, 1, 16385, 10, 1, p, p, None, None, None) == -1
    assert lib.rvs_basis_build(p, None, 1, 100, 33, 1, p, p, None, None, None) == -1
    assert lib.rvs_basis_build(p, None, 1, 100, 0, 1, p, p, None, None, None) == -1
    assert lib.rvs_basis_build(p, None, 1, 100, 10, 1, p, p, p, None, None) == -1


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
             ('rvs_nm_nn_arm', _lib.NmNNArm),
             ('rvs_nm_objective', _lib.NmObjective),
             ('rvs_bfgs_state', _lib.BfgsState),
             ('rvs_tri_buckets', _lib.TriBuckets),
             ('rvs_nm_tri_arm', _lib.NmTriArm)]
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


def _c_prototypes(text):
    """{name: [parameter declarations]} of every `int rvs_*(...)` / `int64_t
    rvs_*(...)` prototype in a C text (comments removed)"""
    import re
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    out = {}
    for m in re.finditer(r'\b(?:int|int64_t)\s+(rvs_\w+)\s*\(([^;{]*?)\)\s*;',
                         text, flags=re.S):
        args = [' '.join(a.split()) for a in m.group(2).split(',')]
        out[m.group(1)] = [] if args == ['void'] else args
    return out


def _norm_type(decl):
    """parameter declaration -> type without the name: 'const double *lam' ->
    'const double*'"""
    import re
    decl = decl.replace('*', ' * ')
    toks = decl.split()
    if toks[-1] != '*' and len(toks) > 1:
        toks = toks[:-1]          # drop the parameter name
    return re.sub(r'\s*\*', '*', ' '.join(toks))


def test_integration_md_prototypes_match_the_header():
    """every C prototype and every ctypes `argtypes` list quoted in
    INTEGRATION.md is the header's (arity and types): a maintainer binds from
    that file"""
    import ctypes
    import re
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(repo, 'INTEGRATION.md')).read()
    hdr = _c_prototypes(open(os.path.join(repo, 'include', 'rvsgpu.h')).read())
    blocks = re.findall(r'```c\n(.*?)```', doc, flags=re.S)
    quoted = {}
    for b in blocks:
        quoted.update(_c_prototypes(b))
    assert {'rvs_spline_construct', 'rvs_spline_eval', 'rvs_chisq_grid'} <= set(quoted)
    for name, args in quoted.items():
        assert name in hdr, name
        assert [_norm_type(a) for a in args] == [_norm_type(a) for a in hdr[name]], name
    # python snippets: L.rvs_x.argtypes = [...]
    P, I = ctypes.c_void_p, ctypes.c_int   # noqa: F841 (names used by eval)
    kinds = {ctypes.c_void_p: 'ptr', ctypes.c_int: 'int', ctypes.c_int64: 'i64',
             ctypes.c_double: 'f64', ctypes.c_uint32: 'u32'}
    n = 0
    for m in re.finditer(r'L\.(rvs_\w+)\.argtypes\s*=\s*(\[.*?\])', doc):
        name, lst = m.group(1), eval(m.group(2))
        want = []
        for a in hdr[name]:
            t = _norm_type(a)
            want.append('ptr' if t.endswith('*') else
                        {'int': 'int', 'int64_t': 'i64', 'double': 'f64',
                         'uint32_t': 'u32'}[t])
        assert [kinds[t] for t in lst] == want, name
        n += 1
    assert n >= 2
    # the calls in the snippets pass as many arguments as the prototype has
    for m in re.finditer(r'L\.(rvs_\w+)\(([^;]*?)\)\n(?:assert|ret|status|rc)',
                         doc, flags=re.S):
        name = m.group(1)
        depth, nargs, cur = 0, 1, ''
        for ch in re.sub(r'#.*', '', m.group(2)):
            if ch in '([':
                depth += 1
            elif ch in ')]':
                depth -= 1
            elif ch == ',' and depth == 0:
                nargs += 1
        assert nargs == len(hdr[name]), (name, nargs, len(hdr[name]))
