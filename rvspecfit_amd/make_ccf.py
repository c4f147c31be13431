"""CCF pre-processing of a spectrum -- API mirror of the part of
py/rvspecfit/make_ccf.py that is on the hot path (what
fitter_ccf.fit calls per spectrum and arm), on the MI355X kernels.

The template side of make_ccf (preprocess_model, ccf_executor, main: building a
library's CCF template set once, offline) is not rebuilt; artefacts made by the
reference are read as they are (library.py, tools/convert_artefacts.py).
get_continuum / fit_resid (the robust continuum of a spectrum, make_ccf.py:105-164)
exist only inside preprocess_data's kernel (rvs_ccf_preprocess: the Levenberg-
Marquardt fit on the device; engine.ccf_preprocess(details=True) returns the
continuum and the node values).
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


def to_power_two(i):
    """make_ccf.to_power_two (make_ccf.py:496-497): the next power of two >= i"""
    return 2**(int(np.ceil(np.log(i) / np.log(2))))


def interp_masker(lam, spec, badmask):
    """make_ccf.interp_masker (make_ccf.py:288-327): the spectrum with its masked
    pixels filled -- linearly in wavelength between the nearest good neighbours, with
    the nearest good value beyond the first / last good pixel; everything masked: the
    spectrum itself with non-finite values set to 1.  Host arithmetic on one spectrum
    (index work; preprocess_data does the same inside its kernel for a batch)."""
    lam = np.asarray(lam)
    spec = np.asarray(spec)
    bad = np.asarray(badmask, dtype=bool)
    out = spec * 1
    good_ix = np.flatnonzero(~bad)
    if good_ix.size == 0:
        import logging
        logging.warning('All the pixels are masked for the ccf determination')
        out[~np.isfinite(out)] = 1
        return out
    bad_ix = np.flatnonzero(bad)
    nxt = np.searchsorted(good_ix, bad_ix)      # first good pixel behind each bad one
    left, right = nxt == 0, nxt == good_ix.size
    inner = ~(left | right)
    out[bad_ix[left]] = spec[good_ix[0]]
    out[bad_ix[right]] = spec[good_ix[-1]]
    a, b = good_ix[nxt[inner] - 1], good_ix[nxt[inner]]
    l0 = lam[bad_ix[inner]]
    out[bad_ix[inner]] = (-(lam[a] - l0) * spec[b] + (lam[b] - l0) * spec[a]) / \
        (lam[b] - lam[a])
    return out


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
