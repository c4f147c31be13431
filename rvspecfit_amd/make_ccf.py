"""CCF pre-processing of a spectrum -- API mirror of the part of
py/rvspecfit/make_ccf.py that is on the hot path (what
fitter_ccf.fit calls per spectrum and arm), on the MI355X kernels.

The template side of make_ccf (preprocess_model, ccf_executor, main: building a
library's CCF template set once, offline) is not rebuilt; artefacts made by the
reference are read as they are (library.py, tools/convert_artefacts.py).
"""
import types

import numpy as np
import torch

from . import _lib
from . import engine


def get_continuum_prefix(continuum):
    """make_ccf.py:19-24"""
    return '' if continuum else 'nocont_'


def get_ccf_info_name(spec_setup, continuum=True):
    return 'ccf_' + get_continuum_prefix(continuum) + '%s.h5' % spec_setup


def get_ccf_dat_name(spec_setup, continuum=True):
    return 'ccfdat_' + get_continuum_prefix(continuum) + '%s.npz' % spec_setup


def get_ccf_mod_name(spec_setup, continuum=True):
    return 'ccfmod_' + get_continuum_prefix(continuum) + '%s.npy' % spec_setup


def get_ccf_config(logl0=None, logl1=None, npoints=None, splinestep=1000,
                   maxcontpts=20):
    """make_ccf.get_ccf_config (make_ccf.py:66-102): the dictionary that describes
    a cross-correlation set-up -- FFT grid log(lambda) logl0..logl1 in `npoints`
    steps, continuum nodes every `splinestep` km/s but at most `maxcontpts` of them
    (splinestep None: no continuum normalisation)."""
    conf = dict(logl0=logl0, logl1=logl1, npoints=npoints, continuum=True,
                maxcontpts=maxcontpts)
    if splinestep is None:
        conf['continuum'] = False
    else:
        widest = 3e5 * (np.exp((logl1 - logl0) / maxcontpts) - 1)
        conf['splinestep'] = max(splinestep, widest)
    return conf


class _ConfLib:
    """what ArmData.ccf_tables reads of a template library: the CCF set-up"""

    def __init__(self, ccfconf):
        self.name = 'ccfconf'
        self._cc = dict(ccfconf)
        self._cc.setdefault('maxcontpts', 20)

    def ccf_set(self, config):
        return self._cc


def preprocess_data(lam, spec0, espec, ccfconf=None, badmask=None, maxerr=10):
    """make_ccf.preprocess_data (make_ccf.py:330-414): the spectrum as the CCF
    templates were prepared -- median filter and masks (error above maxerr medians,
    non-positive filtered flux, the caller's badmask), gaps filled linearly, the
    robust continuum exp(spline) divided out, rebinned onto the FFT grid with the
    inverse variance of the rebinned pixel.  Returns (spec [npoints], ivar
    [npoints]) as numpy; 2-D spec0 / espec / badmask [S, npix] give [S, npoints]
    device tensors.  One block per spectrum of rvs_ccf_preprocess (csrc/ccf.hip)."""
    _lib.require_gpu()
    single = np.ndim(spec0) == 1
    arm = engine.ArmData('ccfconf', np.asarray(lam, dtype=np.float64), spec0,
                         espec, badmask)
    pre = engine.ccf_preprocess(arm, _ConfLib(ccfconf), {}, maxerr=float(maxerr))
    if single:
        return (pre['proc_spec'][0].cpu().numpy(),
                pre['proc_ivar'][0].cpu().numpy())
    return pre['proc_spec'], pre['proc_ivar']
