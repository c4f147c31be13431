"""rvspecfit_amd -- MI355X-native implementation of the rvspecfit likelihood hot
path (template evaluation -> vsini broadening -> Doppler resample ->
continuum-marginalised chi^2 -> RV grid -> FFT cross-correlation).

Module names mirror the reference package (spec_fit, fitter_ccf, vel_fit,
spec_inter, utils) so survey drivers can switch imports.  All numerical work is
done by hand-written HIP kernels behind the C-ABI in include/rvsgpu.h; there is
no CPU fallback in the product path.
"""
__version__ = '0.1.0'
