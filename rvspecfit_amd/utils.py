"""Configuration helpers with the reference's keys and defaults
(py/rvspecfit/utils.py:9-110): yaml -> defaults merge -> hashable frozen dict; and the
driver's file source, FileQueue (utils.py:113-177)."""
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


class FileQueue:
    """utils.FileQueue (utils.py:113-177): the driver's source of file names -- a list,
    the lines of a text file read once, or (queue=True) that text file as a queue
    shared by every process that names it: a process takes the file for itself by
    RENAMING it, removes the first line and renames it back, so two processes never
    get the same line, and GPU ranks (or this build's processes next to the
    reference's CPU workers: same protocol) balance themselves over an uneven file
    list instead of striding it.  `shared` tells proc_many not to stride."""

    def __init__(self, file_list=None, file_from=None, queue=False,
                 retries=1000, wait=(1.0, 1.5)):
        self.shared = False
        self._list, self._path = None, None
        self._retries, self._wait = retries, wait
        if file_list is not None:
            self._list = list(file_list)
        elif file_from is not None and not queue:
            with open(file_from) as fp:
                self._list = [line.rstrip() for line in fp]
        elif file_from is not None:
            self._path = file_from
            self.shared = True
        else:
            raise ValueError('FileQueue needs file_list or file_from')

    def __iter__(self):
        return self

    def __next__(self):
        if self._list is not None:
            if not self._list:
                raise StopIteration
            return self._list.pop(0)
        return self._take()

    def _take(self):
        import random
        import socket
        import time
        mine = '%s.%s.%d.lock' % (self._path, socket.gethostname(), os.getpid())
        for _ in range(self._retries):
            try:
                os.rename(self._path, mine)
            except FileNotFoundError:   # another process holds it
                time.sleep(random.uniform(*self._wait))
                continue
            try:
                with open(mine) as fp:
                    lines = fp.readlines()
                if not lines:
                    raise StopIteration
                with open(mine, 'w') as fp:
                    fp.writelines(lines[1:])
                return lines[0].rstrip()
            finally:
                os.rename(mine, self._path)
        logging.warning('Cannot read next file due to lock')
        raise StopIteration
