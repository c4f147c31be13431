"""Configuration helpers with the reference's keys and defaults
(py/rvspecfit/utils.py:9-110).  Only what the hot path reads is kept: yaml ->
defaults merge -> hashable frozen dict."""
import logging
import os

import yaml


class frozendict(dict):
    """Hashable, immutable dict (stand-in for rvspecfit.frozendict)."""

    def _ro(self, *a, **k):
        raise TypeError('frozendict is immutable')

    __setitem__ = __delitem__ = clear = pop = popitem = setdefault = update = _ro

    def __hash__(self):
        return hash(tuple(sorted((k, _hashable(v)) for k, v in self.items())))


def _hashable(v):
    if isinstance(v, dict):
        return frozendict(v)
    if isinstance(v, list):
        return tuple(v)
    return v


def freezeDict(d):
    if isinstance(d, dict):
        return frozendict({k: freezeDict(v) for k, v in d.items()})
    if isinstance(d, list):
        return tuple(d)
    return d


def get_default_config():
    return dict(min_vel=-1000, max_vel=1000, vel_step0=5, max_vsini=500,
                min_vsini=1e-2, min_vel_step=0.2, second_minimizer=True,
                template_lib='templ_data/')


def read_config(fname=None, override_options=None):
    fname_specified = fname is not None
    if fname is None:
        fname = 'config.yaml'
    if os.path.exists(fname):
        with open(fname, 'r') as fp:
            D = yaml.safe_load(fp) or {}
    else:
        if fname_specified:
            raise RuntimeError(f"Configuration file '{fname}' not found.")
        logging.warning(f"Configuration file '{fname}' not found. "
                        "Using default settings")
        D = {}
    for k, v in get_default_config().items():
        D.setdefault(k, v)
    D['config_file_path'] = os.path.abspath(fname)
    if override_options is not None:
        D.update(override_options)
    return freezeDict(D)
