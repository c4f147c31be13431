"""rvspecfit_amd.plugin: the `interpolation_type = 'generic'` evaluator classes
(spec_inter.py:371-378) that put the MI355X template kernels under an unmodified
rvspecfit."""
import os

import numpy as np
import pytest

from conftest import GOLD, gold_lib_dict
from oracle import rvs_oracle as orc

# what spec_inter.getInterpolator reads from interp_<setup>.h5 for a 'generic'
# record (spec_inter.py:330-340, 371-380)
HOOK_KEYS = ('interpolation_type', 'module', 'class_name', 'outside_class_name',
             'mapper_module', 'mapper_class_name', 'mapper_args', 'parnames',
             'lam', 'log_step', 'log_spec', 'revision')


def test_record_has_what_the_hook_reads():
    import importlib
    from rvspecfit_amd import plugin
    r = plugin.record('gold_b', os.path.join(GOLD, 'lib_gold_b.npz'))
    assert all(k in r for k in HOOK_KEYS)
    assert r['interpolation_type'] == 'generic'
    mod = importlib.import_module(r['module'])
    assert callable(getattr(mod, r['class_name']))
    assert callable(getattr(mod, r['outside_class_name']))
    # LogParamMapper(log_ids) (read_grid.py:104-145)
    assert r['mapper_class_name'] == 'LogParamMapper' and r['mapper_args'] == ([0], )
    d = np.load(os.path.join(GOLD, 'lib_gold_b.npz'))
    np.testing.assert_array_equal(r['lam'], d['lam'])
    assert r['parnames'] == ('teff', 'logg', 'feh', 'alpha')
    nn = dict(np.load(os.path.join(GOLD, 'nn_case.npz')))


def test_no_cpu_fallback():
    """without a GPU the constructor raises (the product never evaluates
    templates on the CPU)"""
    import torch
    from rvspecfit_amd import plugin
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    fd = dict(plugin.record('gold_b', os.path.join(GOLD, 'lib_gold_b.npz')),
              template_lib=GOLD)
    with pytest.raises(RuntimeError):
        plugin.Evaluator(fd)


@pytest.mark.gpu
def test_plugin_evaluator_and_outside(tmp_path, cases):
    """Evaluator(fd)(mapper.forward(p)) / Outside(fd)(...) -- the calls
    SpecInterpolator.eval / .outsideFlag make (spec_inter.py:257-286) -- against
    the reference's captured GridInterp / GridOutsideCheck outputs and the
    oracle, in-grid, on a hole and outside the grid"""
    from rvspecfit_amd import plugin
    for name in ('gold_b', 'gold_r'):
        np.savez(os.path.join(tmp_path, 'rvsgpu_%s.npz' % name),
                 **gold_lib_dict(name))
        fd = plugin.record(name, os.path.join(tmp_path, 'rvsgpu_%s.npz' % name))
        fd['template_lib'] = str(tmp_path)     # added by getInterpolator (:376)
        ev, out = plugin.Evaluator(fd), plugin.Outside(fd)
        assert ev.lib is out.lib               # one device copy per artefact
        olib = orc.Library(gold_lib_dict(name))
        P = cases['interp/params']
        for i, p in enumerate(P):
            mp = olib.map_params(p)            # LogParamMapper.forward
            t = ev(mp)
            assert t.dtype == np.float64 and t.shape == (len(olib.lam), )
            ref = cases['interp/%s/eval' % name][i]
            nearest = olib.eval(p, details=True)[1]['nearest'] >= 0
            np.testing.assert_allclose(t, ref, rtol=3e-7 if nearest else 1e-12)
            oref = cases['interp/%s/outside' % name][i]
            o = out(mp)
            if np.isfinite(oref):
                assert abs(o - oref) <= 1e-12 * max(1, abs(oref))
            else:
                assert not np.isfinite(o)
        tb, ob = ev.batch(np.array([olib.map_params(p) for p in P]))
        np.testing.assert_array_equal(tb[2].cpu().numpy(), ev(olib.map_params(P[2])))
    with pytest.raises(ValueError):
        ev([3.7, 2.0])
    with pytest.raises(RuntimeError):
        plugin.Evaluator(dict(fd, rvsgpu_file='missing.npz'))


@pytest.mark.gpu
def test_plugin_nn_setup(tmp_path):
    """an MLP setup behind the same hook: the input is nn Mapper.forward(p)
    (nn/NNInterpolator.py:159-171), the output what RVSInterpolator.__call__ /
    OutsideInterpolator.__call__ return (nn/RVSInterpolator.py:36-71)"""
    from rvspecfit_amd import plugin
    d = dict(np.load(os.path.join(GOLD, 'nn_case.npz')))
    lam = np.exp(np.linspace(np.log(4000.), np.log(4100.), int(d['dims'][-1])))
    dd = dict(lam=lam, log_step=np.array(True), log_ids=np.array([0]),
              parnames=np.array(['teff', 'logg', 'feh', 'alpha']),
              nn_dims=d['dims'], nn_M=d['M'], nn_S=d['S'], nn_pts=d['pts'])
    for i in range(len(d['dims']) - 1):
        dd['nn_W%d' % i] = d['W%d' % i]
        dd['nn_b%d' % i] = d['b%d' % i]
    f = os.path.join(tmp_path, 'rvsgpu_nn_test.npz')
    np.savez(f, **dd)
    fd = plugin.record('nn_test', f, mapper_module='rvspecfit.nn.NNInterpolator',
                       mapper_class_name='Mapper')
    fd['template_lib'] = str(tmp_path)
    ev, out = plugin.Evaluator(fd), plugin.Outside(fd)
    for p, t_ref, o_ref in zip(d['params'], d['out'], d['outside']):
        x1 = np.asarray(p, dtype=np.float32)   # Mapper.forward
        y = x1 * 1
        y[0] = np.log10(x1[0])
        mp = (y - d['M']) / d['S']
        np.testing.assert_allclose(ev(mp), t_ref, rtol=3e-6)
        assert abs(out(mp) - o_ref) <= 1e-5 * abs(o_ref) + 1e-9
