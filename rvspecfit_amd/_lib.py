"""ctypes binding of librvsgpu.so (the C-ABI declared in include/rvsgpu.h).

There is NO fallback: if the shared library is missing the import of any
compute entry point raises.  Device pointers are taken from torch tensors
(`tensor.data_ptr()`); the stream is torch's current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'librvsgpu.so')

P = ctypes.c_void_p
I = ctypes.c_int
L = ctypes.c_int64
D = ctypes.c_double
U = ctypes.c_uint32

# name -> (restype, argtypes); mirrors include/rvsgpu.h declaration by declaration
SIGNATURES = {
    'rvs_abi_version': (I, []),
    'rvs_option_set': (I, [ctypes.c_char_p, I]),
    'rvs_option_get': (I, [ctypes.c_char_p, P]),
    'rvs_template_polylinear': (I, [P, L, I, P, P, P, I, P, P, U, I, P, I, P, P,
                                    P, P, P]),
    'rvs_template_tri': (I, [P, I, P, P, P, I, I, U, I, P, I, P, P, P, P, P]),
    'rvs_template_tri_buckets': (I, [P, I, P, P, P, I, I, U, I, P, P, I, P, P, P,
                                     P, P]),
    'rvs_vsini_convolve': (I, [P, P, P, D, D, I, I, P, P]),
    'rvs_spline_factors': (I, [P, I, P, P]),
    'rvs_spline_factors_len': (ctypes.c_int64, [I]),
    'rvs_spline_construct': (I, [P, P, I, I, I, P, P, P]),
    'rvs_spline_eval': (I, [P, P, I, I, P, I, I, P, P, P, P]),
    'rvs_chisq_work_size': (L, [I, I]),
    'rvs_chisq_work_size_g': (L, [I, I, I]),
    'rvs_chisq_prepare_g': (I, [P, P, P, I, I, I, P, I, D, P, P]),
    'rvs_chisq_grid_g': (I, [P, P, P, I, I, I, P, I, L, P, P, I, I, I, P, P, I, P, L,
                             I, P, D, D, I, P, P, P, P]),
    'rvs_chisq_prepare': (I, [P, P, P, I, I, P, I, D, P, P]),
    'rvs_chisq_grid': (I, [P, P, P, I, I, I, P, P, I, I, I, P, P, I, P, L, I, P,
                           D, D, I, P, P, P]),
    'rvs_chisq_grid_resol': (I, [P, P, P, I, I, I, P, P, I, I, I, P, I, L, P, P,
                                 I, P, L, I, P, D, D, P, P, P]),
    'rvs_chisq_grid_resol_g': (I, [P, P, P, I, I, I, P, I, L, P, P, I, I, I, P, I, L,
                                   P, P, I, P, L, I, P, D, D, P, P, P, P]),
    'rvs_chisq_full': (I, [P, P, P, P, P, I, I, I, P, P, I, I, I, I, I, P, P, I,
                           P, D, I, P, I, L, P, P, P, P, P, P, P, P]),
    'rvs_chisq_full_g': (I, [P, P, P, P, P, I, I, I, P, P, I, I, I, I, I, P, P, I,
                             P, D, I, P, I, L, P, P, P, P, P, P, P, P, I, L, P]),
    'rvs_chisq_continuum_work_size': (L, [I, I]),
    'rvs_chisq_continuum_g': (I, [P, P, P, P, P, I, I, I, P, P, P, P, P, P, I, L,
                                  P]),
    'rvs_chisq_continuum': (I, [P, P, P, P, P, I, I, I, P, P, P, P, P, P]),
    'rvs_chisq_point_work_size': (L, [I, I]),
    'rvs_chisq_point': (I, [P, I, I, P, P, I, P, D, P, P, P, P]),
    'rvs_nm_begin': (I, [I, I, D, D, I, P, P, P, P, P, P, P, I, P]),
    'rvs_nm_decide': (I, [I, P, P, P, P, P, P, P, P, P, I, P]),
    'rvs_nm_update': (I, [I, P, P, P, P, P, P, P, P, P, P, P, P, P, I, P]),
    'rvs_nm_collect': (I, [I, P, P, P, P]),
    'rvs_nm_shrink_point': (I, [I, I, P, P, P, P, I, P]),
    'rvs_nm_shrink_store': (I, [I, I, P, P, P, P, P, P, P, P, I, P]),
    'rvs_proc_map': (I, [I, I, I, P, P, P, I, P, P, P, P, P, D, D, D, P, P, P, P,
                          P, P, P]),
    'rvs_proc_finish': (I, [I, P, I, P, P, P, P, P, P, P, P]),
    'rvs_nm_run': (I, [P, P, D, D, I, I, P, P]),
    'rvs_bfgs_begin': (P, [I, I, P, P, D, D, D, D, I]),
    'rvs_bfgs_pending': (L, [P, P, P, L]),
    'rvs_bfgs_feed': (I, [P, P, L]),
    'rvs_bfgs_result': (I, [P, P, P, P, P, P, P, P]),
    'rvs_bfgs_end': (None, [P]),
    'rvs_bfgs_run_bytes': (L, []),
    'rvs_bfgs_run': (I, [P, P, I, P, P]),
    'rvs_objective_max_ntp': (I, [I]),
    'rvs_objective_work_size': (L, [I, I]),
    'rvs_objective_fused': (I, [P, I, I, P, P, P, I, P, D, I, P, P, P, P]),
    'rvs_objective_from_template': (I, [P, I, I, P, P, P, P, I, P, D, I, P, P, P,
                                        P]),
    'rvs_objective_fused_n': (I, [P, I, I, P, P, P, I, P, P, D, I, P, P, P, P]),
    'rvs_objective_from_template_n': (I, [P, I, I, P, P, P, P, I, P, P, D, I, P,
                                          P, P, P]),
    'rvs_grid_moments': (I, [P, P, L, P, I, I, I, I, P, P, P, P]),
    'rvs_basis_build': (I, [P, P, I, I, I, I, P, P, P, P, P]),
    'rvs_ccf_tables_build': (I, [P, P, I, I, P, I, I, P, P, P, I, P, P, P, P, P, P,
                                 P]),
    'rvs_ccf_preprocess': (I, [P, P, P, P, I, I, I, P, P, P, P, I, P, P, P, I, D,
                               P, P, P, P, P, P, P]),
    'rvs_ccf_preprocess_g': (I, [P, P, P, P, I, I, I, P, P, P, P, I, P, P, P, I,
                                 D, P, P, P, P, P, P, P, P, P, P]),
    'rvs_ccf_fft_pos': (I, [I, I]),
    'rvs_ccf_xcorr': (I, [P, P, I, I, P, P, I, P, I, P, P, I, P, P, I, D, P, P,
                          P, P]),
    'rvs_ccf_select': (I, [P, P, I, I, I, P, I, P, P, P, P]),
    'rvs_template_nn': (I, [P, I, I, U, P, P, I, P, P, P, P, P, P, P]),
    'rvs_template_nn_arms': (I, [P, I, I, I, P, P]),
    'rvs_template_nn_arms_n': (I, [P, I, P, I, I, P, P]),
    'rvs_nn_outside': (I, [P, I, I, U, P, P, I, P, I, P, I, P, P]),
}

_lib = None
# RVS_ABI_VERSION of the include/rvsgpu.h these signatures mirror
ABI_VERSION = 12


class RvsGpuError(RuntimeError):
    pass


def lib():
    """Load librvsgpu.so; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RvsGpuError(
                '%s not found: build it with `make -C rvspecfit_amd/csrc` '
                '(or python -c "import __graft_entry__ as g; g.build()"). '
                'rvspecfit_amd has no CPU fallback.' % LIB_PATH)
        L_ = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L_, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if L_.rvs_abi_version() != ABI_VERSION:
            raise RvsGpuError(
                '%s is ABI version %d, this package binds version %d: rebuild it '
                '(make -C rvspecfit_amd/csrc)' % (LIB_PATH, L_.rvs_abi_version(),
                                                   ABI_VERSION))
        _lib = L_
    return _lib


def set_option(name, value):
    """rvs_option_set; returns the previous value"""
    old = ctypes.c_int(0)
    check(lib().rvs_option_get(name.encode(), ctypes.byref(old)), 'rvs_option_get')
    check(lib().rvs_option_set(name.encode(), int(value)), 'rvs_option_set')
    return old.value


class option:
    """`with _lib.option('xc_ws', 0): ...` -- a switch for the duration of a block
    (tests holding one kernel path against another)"""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


class NmState(ctypes.Structure):
    """rvs_nm_state of include/rvsgpu.h"""
    _fields_ = [(k, ctypes.c_void_p) for k in
                ('sim', 'fsim', 'X1', 'X2', 'F1', 'F2', 'nit', 'nfev', 'flags',
                 'list1', 'list2', 'list3', 'cases', 'pos2', 'counts')] + [
                    ('S', ctypes.c_int32), ('N', ctypes.c_int32),
                    ('stop_below', ctypes.c_int32), ('reserved_', ctypes.c_int32)]


class NmNNArm(ctypes.Structure):
    """rvs_nm_nn_arm of include/rvsgpu.h"""
    _fields_ = [(k, ctypes.c_void_p) for k in
                ('M', 'S', 'W', 'b', 'dims', 'act0', 'act1', 'templ', 'outside',
                 'xeqs', 'yeqs')] + [(k, ctypes.c_int32) for k in
                                     ('nlayer', 'nfx', 'nfy')] + [
                    ('log_mask', ctypes.c_uint32)]


class NmObjective(ctypes.Structure):
    """rvs_nm_objective of include/rvsgpu.h"""
    _fields_ = [(k, ctypes.c_void_p) for k in
                ('arms', 'fixed', 'vsini_fixed', 'safe', 'prior_mean',
                 'prior_isig', 'vel', 'vsini', 'params', 'extra', 'chi',
                 'job_spec', 'bad', 'jstatus', 'status', 'scratch')] + [
                    (k, ctypes.c_double) for k in
                    ('min_vel', 'max_vel', 'max_vsini', 'badchi')] + [
                    (k, ctypes.c_int32) for k in
                    ('narm', 'npoly', 'n', 'ndim', 'vsini_col')] + [
                    ('src', ctypes.c_int32 * 8), ('nn', ctypes.c_void_p),
                    ('tri', ctypes.c_void_p)]


class BfgsState(ctypes.Structure):
    """rvs_bfgs_state of include/rvsgpu.h"""
    _fields_ = [(k, ctypes.c_void_p) for k in
                ('runs', 'x0', 'hess_inv0', 'x', 'fun', 'hess_inv', 'nit', 'nfev',
                 'status', 'nreq', 'off', 'list', 'counts', 'X', 'F')] + [
                    (k, ctypes.c_double) for k in
                    ('gtol', 'c1', 'c2', 'xrtol')] + [
                    (k, ctypes.c_int32) for k in ('S', 'n', 'cap', 'maxiter')]


class TriBuckets(ctypes.Structure):
    """rvs_tri_buckets of include/rvsgpu.h"""
    _fields_ = [('cell_start', ctypes.c_void_p), ('cell_list', ctypes.c_void_p),
                ('lo', ctypes.c_double * 6), ('inv_w', ctypes.c_double * 6),
                ('n', ctypes.c_int32 * 6)]


class NmTriArm(ctypes.Structure):
    """rvs_nm_tri_arm of include/rvsgpu.h"""
    _fields_ = [(k, ctypes.c_void_p) for k in
                ('dats', 'transform', 'extraflags', 'simplices', 'templ', 'outside',
                 'simplex')] + [('buckets', TriBuckets)] + [
                    (k, ctypes.c_int32) for k in ('ntp', 'nsimplex', 'exp_flag')] + [
                    ('log_mask', ctypes.c_uint32)]


class PointArm(ctypes.Structure):
    """rvs_point_arm of include/rvsgpu.h"""
    _fields_ = [(k, ctypes.c_void_p) for k in
                ('lam', 'polysT', 'spec', 'espec', 'work', 'knots', 'coef',
                 'penalty', 'taps')] + [('taps_stride', ctypes.c_int64),
                                        ('espec_sys', ctypes.c_double)] + [
                    (k, ctypes.c_int32) for k in
                    ('npix', 'S', 'ntp', 'log_step', 'nd', 'fast_interp')] + [
                    ('grid_id', ctypes.c_void_p), ('polys_stride', ctypes.c_int64),
                    ('G', ctypes.c_int32), ('reserved_', ctypes.c_int32),
                    ('pen_scale', ctypes.c_void_p)]


class ObjectiveArm(ctypes.Structure):
    """rvs_objective_arm of include/rvsgpu.h"""
    _fields_ = [('pt', PointArm), ('dats', ctypes.c_void_p),
                ('idgrid', ctypes.c_void_p), ('uvecs', ctypes.c_void_p),
                ('vecs_s', ctypes.c_void_p), ('factors', ctypes.c_void_p),
                ('ngrid', ctypes.c_int64), ('lnstep', ctypes.c_double),
                ('ptp', ctypes.c_double * 6), ('lens', ctypes.c_int32 * 6),
                ('ntp', ctypes.c_int32), ('ndim', ctypes.c_int32),
                ('log_mask', ctypes.c_uint32), ('exp_flag', ctypes.c_int32)]


def ptr(t):
    """device (or host, for numpy/ctypes arrays) pointer of a tensor, or None"""
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        assert t.is_contiguous(), 'tensor must be contiguous'
        return ctypes.c_void_p(t.data_ptr())
    return ctypes.c_void_p(t.ctypes.data)  # numpy (host pointers)


def stream():
    if torch.cuda.is_available():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    return None


def check(rc, what):
    if rc != 0:
        raise RvsGpuError('%s failed with code %d' % (what, rc))


def require_gpu():
    if not torch.cuda.is_available():
        raise RvsGpuError('rvspecfit_amd needs a ROCm GPU (MI355X); '
                          'there is no CPU path')


# status bits (include/rvsgpu.h)
ST_SPLINE_RANGE = 0x1
ST_SPLINE_GRID = 0x2
ST_NONFINITE = 0x4
ST_CHOL_FALLBACK = 0x8
ST_OUTSIDE_NAN = 0x10
ST_CCF_FAILED = 0x20
ST_ALLMASKED = 0x40
ST_QUAD_ASSERT = 0x80
ST_ILLCOND = 0x100
