import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session', autouse=True)
def _build_oracle():
    so = os.path.join(REPO, 'oracle', '_build', 'liboracle_core.so')
    if not os.path.exists(so):
        subprocess.check_call(['make', '-C', os.path.join(REPO, 'oracle')],
                              stdout=subprocess.DEVNULL)


@pytest.fixture(scope='session')
def cases():
    return dict(np.load(os.path.join(GOLD, 'cases.npz')))


def gold_lib_dict(n):
    """converted artefact of a golden setup: interpolation data + both CCF
    template sets (continuum-normalised ccf_*, non-normalised ccfnc_*)"""
    d = dict(np.load(os.path.join(GOLD, 'lib_%s.npz' % n)))
    d.update(np.load(os.path.join(GOLD, 'lib_nocont_%s.npz' % n)))
    return d


@pytest.fixture(scope='session')
def gold_libs():
    from oracle import rvs_oracle as orc
    return {n: orc.Library(gold_lib_dict(n)) for n in ('gold_b', 'gold_r')}


GOLD_CONFIG = dict(min_vel=-1000, max_vel=1000, min_vel_step=0.2, vel_step0=5,
                   min_vsini=0.1, max_vsini=500)


@pytest.fixture(scope='session')
def gold_config():
    return dict(GOLD_CONFIG)


def gold_specdata(cases, tag, cls):
    names = [str(_) for _ in cases[tag + '/names']]
    return [
        cls(n, cases['%s/%s/lam' % (tag, n)], cases['%s/%s/spec' % (tag, n)],
            cases['%s/%s/espec' % (tag, n)],
            badmask=cases['%s/%s/badmask' % (tag, n)]) for n in names
    ]
