#!/usr/bin/env python
"""Convert an rvspecfit template directory into the .npz files the MI355X
engine loads (rvspecfit_amd.library.TemplateLibrary / spec_inter.getInterpolator).

    <python with h5py>  tools/convert_artefacts.py TEMPLATE_DIR SETUP [SETUP ...]
    <python with torch> tools/convert_artefacts.py TEMPLATE_DIR SETUP --nn-weights

Reads the reference's on-disk formats WITHOUT importing the reference:
  interp_<setup>.h5      dict<->HDF5 schema of rvspecfit/serializer.py:10-169
                         (every dataset carries a 'type' attribute: scalar, str,
                         ndarray, list, tuple, None, flattened_list/_tuple with
                         __item_<i> members)
  interpdat_<setup>.npy  float32 [N_grid, n_tpix] log-flux (make_nd.py:176)
  ccf_<setup>.h5, ccfdat_<setup>.npz, ccfmod_<setup>.npy  (make_ccf.py:483-493)
  NNstate_<setup>.sav    torch checkpoint wrapper (nn/NNInterpolator.py:8-156)
and writes TEMPLATE_DIR/rvsgpu_<setup>.npz.  h5py is not installed next to
torch in the build image, hence the two-pass option for NN libraries.
"""
import argparse
import os
import sys

import numpy as np


def _h5_to_dict(group):
    import h5py
    out = {}
    for key, item in group.items():
        typ = item.attrs.get('type', None)
        if isinstance(typ, bytes):
            typ = typ.decode()
        if isinstance(item, h5py.Group):
            sub = _h5_to_dict(item)
            if typ in ('flattened_list', 'flattened_tuple'):
                n = len(sub)
                seq = [sub['__item_%d' % i] for i in range(n)]
                out[key] = seq if typ.endswith('list') else tuple(seq)
            else:
                out[key] = sub
            continue
        val = item[()]
        if typ == 'None':
            out[key] = None
        elif typ == 'str':
            out[key] = val.decode() if isinstance(val, bytes) else str(val)
        elif typ in ('list', 'tuple', 'ndarray', 'empty_array'):
            arr = np.asarray(val)
            if arr.dtype.kind in ('O', 'S'):
                arr = np.array([_.decode() if isinstance(_, bytes) else _
                                for _ in arr.ravel()]).reshape(arr.shape)
            out[key] = arr if typ == 'ndarray' else (
                list(arr) if typ in ('list', 'empty_array') else tuple(arr))
        elif typ == 'pickle':
            # the serializer's escape hatch (serializer.py): a pickled object in
            # a uint8 dataset -- used for the scipy.spatial.Delaunay object of
            # triangulation libraries (make_nd.py:137-138, 174-175)
            import pickle
            try:
                out[key] = pickle.loads(np.asarray(val).tobytes())
            except Exception:
                out[key] = None
        else:
            out[key] = val.item() if hasattr(val, 'item') and np.ndim(val) == 0 \
                else val
    return out


def load_h5(fname):
    import h5py
    with h5py.File(fname, 'r') as fp:
        return _h5_to_dict(fp)


def convert(tdir, setup):
    fd = load_h5(os.path.join(tdir, 'interp_%s.h5' % setup))
    out = dict(lam=np.asarray(fd['lam'], dtype=np.float64),
               log_step=np.array(bool(fd['log_step'])),
               parnames=np.array([str(_) for _ in fd['parnames']]),
               revision=np.array(str(fd.get('revision') or '')),
               creation_soft_version=np.array(
                   str(fd.get('creation_soft_version') or '')))
    itype = fd.get('interpolation_type')
    if itype is None:
        itype = 'regulargrid' if 'regular' in fd else 'triangulation'
    if itype == 'regulargrid':
        out['log_ids'] = np.asarray(fd['mapper_args'][0], dtype=np.int64).ravel()
        out['dats'] = np.load(os.path.join(tdir, 'interpdat_%s.npy' % setup))
        out['vec'] = np.asarray(fd['vec'], dtype=np.float64)
        out['idgrid'] = np.asarray(fd['idgrid'], dtype=np.int64)
        out['log_spec'] = np.array(bool(fd.get('log_spec', True)))
        for i, u in enumerate(fd['uvecs']):
            out['uvec%d' % i] = np.asarray(u, dtype=np.float64)
    elif itype == 'generic':
        M, S, log_ids = fd['mapper_args']
        out['log_ids'] = np.asarray(log_ids, dtype=np.int64).ravel()
        out['nn_M'] = np.asarray(M, dtype=np.float64)
        out['nn_S'] = np.asarray(S, dtype=np.float64)
        out['nn_pts'] = np.asarray(fd['outside_kwargs']['pts'], dtype=np.float64)
        kw = fd['class_kwargs']
        out['nn_dims'] = np.array([kw['indim']] + [kw['width']] *
                                  (kw['nlayers'] + 1) + [kw['npc'], kw['npix']],
                                  dtype=np.int32)
        out['nn_file'] = np.array(str(fd['nn_file']))
    elif itype == 'triangulation':
        tri = fd.get('triang')
        if tri is None:
            raise SystemExit('cannot unpickle the Delaunay object of %s with '
                             'this scipy; run the converter under the '
                             'interpreter that wrote it' % setup)
        out['log_ids'] = np.asarray(fd['mapper_args'][0], dtype=np.int64).ravel()
        out['dats'] = np.load(os.path.join(tdir, 'interpdat_%s.npy' % setup)
                              ).astype(np.float64)
        out['vec'] = np.asarray(fd['vec'], dtype=np.float64)
        out['simplices'] = np.asarray(tri.simplices, dtype=np.int32)
        out['transform'] = np.asarray(tri.transform, dtype=np.float64)
        out['extraflags'] = np.asarray(fd['extraflags'],
                                       dtype=np.float64).reshape(-1)
        out['log_spec'] = np.array(bool(fd.get('log_spec', True)))
    else:
        raise SystemExit('unknown interpolation_type %s' % itype)
    # both CCF template sets of the setup, when present (make_ccf.py:19-36):
    # ccf_<setup>.h5 ... -> keys ccf_*, ccf_nocont_<setup>.h5 ... -> keys ccfnc_*;
    # config['ccf_continuum_normalize'] picks one at run time
    # (fitter_ccf.py:40-47)
    for pref, key in (('', 'ccf_'), ('nocont_', 'ccfnc_')):
        cinfo = os.path.join(tdir, 'ccf_%s%s.h5' % (pref, setup))
        if not os.path.exists(cinfo):
            continue
        ci = load_h5(cinfo)
        cd = np.load(os.path.join(tdir, 'ccfdat_%s%s.npz' % (pref, setup)))
        cc = ci['ccfconf']
        out.update({
            key + 'fft': cd['fft'], key + 'fft2': cd['fft2'],
            key + 'mod': np.load(os.path.join(
                tdir, 'ccfmod_%s%s.npy' % (pref, setup))),
            key + 'params': np.asarray(ci['params'], dtype=np.float64),
            key + 'vsinis': np.array([np.nan if _ is None else float(_)
                                      for _ in ci['vsinis']]),
            key + 'parnames': np.array([str(_) for _ in ci['parnames']]),
            key + 'logl0': np.array(float(cc['logl0'])),
            key + 'logl1': np.array(float(cc['logl1'])),
            key + 'npoints': np.array(int(cc['npoints'])),
            key + 'continuum': np.array(bool(cc['continuum'])),
            key + 'maxcontpts': np.array(int(cc.get('maxcontpts', 20)))})
        if cc.get('splinestep') is not None:
            out[key + 'splinestep'] = np.array(float(cc['splinestep']))
    ofile = os.path.join(tdir, 'rvsgpu_%s.npz' % setup)
    np.savez(ofile, **out)
    return ofile


def add_nn_weights(tdir, setup):
    """second pass under an interpreter with torch: fold the checkpoint's
    float32 weights into rvsgpu_<setup>.npz"""
    import torch
    ofile = os.path.join(tdir, 'rvsgpu_%s.npz' % setup)
    d = dict(np.load(ofile))
    ck = torch.load(os.path.join(tdir, str(d['nn_file'])), map_location='cpu',
                    weights_only=True)
    if isinstance(ck, dict) and 'state_dict' in ck:
        if ck.get('checkpoint_magic') != 'rvspecfit.nn_interpolator':
            raise SystemExit('not an rvspecfit NN checkpoint')
        ck = ck['state_dict']
    nl = len(d['nn_dims']) - 1
    names = ['model.lin_%d' % i for i in range(nl - 1)] + ['pc_layer']
    for i, k in enumerate(names):
        d['nn_W%d' % i] = ck[k + '.weight'].numpy().astype(np.float32)
        d['nn_b%d' % i] = ck[k + '.bias'].numpy().astype(np.float32)
    np.savez(ofile, **d)
    return ofile


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('template_dir')
    ap.add_argument('setups', nargs='+')
    ap.add_argument('--nn-weights', action='store_true')
    a = ap.parse_args()
    for s in a.setups:
        if a.nn_weights:
            print(add_nn_weights(a.template_dir, s))
        else:
            print(convert(a.template_dir, s))


if __name__ == '__main__':
    main()
