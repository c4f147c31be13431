/*
 * rvsgpu.h -- C-ABI of librvsgpu.so, the MI355X (gfx950) implementation of the
 * rvspecfit likelihood hot path.
 *
 * Conventions (they follow the reference's only C-ABI, py/rvspecfit/ffibuilder.py:10-17,
 * i.e. `construct` / `evaler` of py/rvspecfit/src/spliner.c):
 *   - extern "C", plain pointers and sizes, no torch / C++ types;
 *   - the CALLER owns every buffer; the callee never allocates or retains memory;
 *   - every pointer is a DEVICE pointer (HBM) unless the comment says "host";
 *   - all floating point data is float64 (the reference computes in float64),
 *     except the template grid `dats` and the NN weights which are float32 as in
 *     the reference's artefacts;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     launches are asynchronous, nothing synchronises;
 *   - return value: 0 ok; <0 argument / shape / launch error (RVS_E_*).
 *     Per-spectrum conditions that the reference reports through Python
 *     exceptions are returned as bit flags in an int32 `status[]` output
 *     (RVS_ST_*); the Python layer re-raises the reference's exception types
 *     for batch-of-1 calls.
 *
 * Citations below are relative to /root/reference/py/rvspecfit/.
 */
#ifndef RVSGPU_H
#define RVSGPU_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RVS_E_ARG (-1)     /* bad argument / unsupported shape */
#define RVS_E_LAUNCH (-2)  /* hipLaunch failure (hipGetLastError != 0) */

/* per-job status bits */
#define RVS_ST_SPLINE_RANGE 0x1   /* evaler() == -1 : eval point outside knots (spliner.c:78-83)  */
#define RVS_ST_SPLINE_GRID 0x2    /* evaler() == -2 : knots not uniform (spliner.c:87,95)          */
#define RVS_ST_NONFINITE 0x4      /* non finite -2 log L (spec_fit.py:963-974)                     */
#define RVS_ST_CHOL_FALLBACK 0x8  /* Cholesky failed, eigen (SVD) branch used (spec_fit.py:337-354) */
#define RVS_ST_OUTSIDE_NAN 0x10   /* template outside grid & not finite (spec_fit.py:392-397)      */
#define RVS_ST_CCF_FAILED 0x20    /* non finite CCF minimum (fitter_ccf.py:234-236)                 */
#define RVS_ST_ALLMASKED 0x40     /* every pixel masked in CCF preprocessing (make_ccf.py:311-315)  */
#define RVS_ST_QUAD_ASSERT 0x80   /* parabola vertex outside its bracket (spec_fit.py:1014 assert)  */
#define RVS_ST_ILLCOND 0x100      /* rvs_chisq_grid: normal matrix pivots span > 1e9 (long stretch of weightless pixels); re-evaluate the job with rvs_chisq_point */

/* library version / build probe (host).  RVS_ABI_VERSION changes whenever the
 * meaning of an argument, a status bit or a work-size formula changes; a caller
 * compares rvs_abi_version() with the header it was built against
 * (rvspecfit_amd/_lib.py refuses a stale librvsgpu.so).
 *   2: status bit 0x100 = RVS_ST_ILLCOND; rvs_chisq_grid's int after `beta` is
 *      pack_min_jobs; rvs_chisq_continuum_work_size = 8 doubles per spectrum
 *   3: rvs_template_polylinear / rvs_objective_arm take `ptp` (the query is
 *      p / ptp as in spec_inter.py:130-132), not its reciprocal
 *   4: rvs_objective_work_size grew (the cell-search records of the objective's
 *      locate pass live in the caller's scratch)
 *   5: rvs_nn_outside added; rvs_chisq_grid packs left-over velocities from
 *      2000 jobs up by default (pack_min_jobs = 0)
 *   6: rvs_nm_objective.nn (MLP evaluators inside rvs_nm_run),
 *      rvs_template_nn_arms
 *   7: grid sets -- spectra of one arm on different wavelength grids in one batch
 *      (spec_fit.py:70-145 takes any `lam` per object): rvs_chisq_work_size_g,
 *      rvs_chisq_prepare_g, rvs_chisq_grid_g, rvs_chisq_full_g,
 *      rvs_chisq_continuum_g, rvs_ccf_preprocess_g; rvs_point_arm grew (grid_id,
 *      polys_stride, G); rvs_objective_work_size grew (the jobs' cell order);
 *      rvs_nn_outside accepts nfx = nfy = 0; rvs_basis_build, rvs_ccf_tables_build
 *      (the per-grid tables on the device)
 *   8: rvs_chisq_work_size(_g) grew by 2*S*npix doubles: rvs_chisq_prepare also
 *      leaves {1/e, s/e} per spectrum pixel (e with espec_sys in quadrature), which
 *      rvs_objective_fused / rvs_nm_run read instead of spec / espec;
 *      rvs_objective_fused_n / rvs_objective_from_template_n (job count on the
 *      device)
 *   9: rvs_option_set / rvs_option_get (the behaviour switches, no longer read from
 *      the environment at launch time); rvs_basis_build takes npoly <= 32 and
 *      npix <= 16384; espec = +inf marks padding only when G > 1
 *  10: rvs_chisq_full(_g) takes unit_template = 2 (the template given on the
 *      pixels: get_chisq0 itself); rvs_objective_fused refuses npoly > 10 on a
 *      template of 2 ntp < 8 (npoly (npoly + 3) / 2 + 1) knots
 *  11: rvs_spline_factors writes rvs_spline_factors_len(ntp) doubles (the five arrays
 *      of ntp, then the same factors in the objective kernel's chunk order); the cell
 *      record of rvs_objective_work_size grew by three doubles per (job, arm);
 *      rvs_chisq_work_size(_g) grew by 2*G*npix doubles ({lam, pix} pairs);
 *      options nm_split_min, nm_spec_max, nm_tail_window (additions: no signature
 *      changed);
 *      rvs_nm_run uses counts[5] (rows of a round that evaluates all candidates)
 *  12: rvs_bfgs_run / rvs_bfgs_run_bytes (the second minimiser's rounds on the
 *      device), rvs_chisq_grid_resol_g (resolution matrices on grid sets),
 *      rvs_template_tri_buckets (find_simplex through a bucket grid);
 *      rvs_nm_objective grew by `tri` (Delaunay libraries inside rvs_nm_run),
 *      rvs_nm_state by `stop_below` */
#define RVS_ABI_VERSION 12
int rvs_abi_version(void);

/* ------------------------------------------------------------------------
 * Behaviour switches.  The entry points may be driven from several host threads
 * (vel_fit.process runs its halves on two), so no switch is read from the
 * environment at launch time.  The table is filled once, on first use, from the
 * environment variables RVS_<NAME in upper case>; afterwards only rvs_option_set
 * changes it, and a change takes effect at the next call that looks.  Every
 * setting gives the same results (to the bit, or -- xc_ws -- to a few ulp); the
 * switches exist so that tests can hold one kernel path against another.
 *   "xc_ws"          1  rvs_ccf_xcorr: the wave-specialised persistent kernels
 *                       where they apply; 0 = one block per (spectrum, template)
 *   "xc_ws1"         0  nfft 4096: 1 = one template per iteration (not two)
 *   "nm_glue"        1  rvs_nm_run: a round as three bookkeeping kernels; 0 = the
 *                       chain of stand-alone kernels (rvs_nm_begin / _decide / ...)
 *   "nm_bucket"      0  rvs_nm_run: launch bounds rounded up to buckets
 *   "nm_split_min" 1024 rvs_nm_run: rounds of at least this many rows (and the test
 *                       of all simplices at the start and after a shrink, for at
 *                       least this many simplices) run their bookkeeping as a
 *                       row-parallel kernel + a one-block pack
 *   "nm_spec_max"   21  rvs_nm_run: rounds of at most this many rows (<= 64, and a
 *                       quarter of the simplices) evaluate all four candidate points
 *                       of a step in one launch, with one bookkeeping kernel per
 *                       round (same state, bit for bit); 0 = never
 *   "nm_tail_window" 16 rvs_nm_run: rounds between two looks of the host (counter
 *                       copy + stream synchronisation) once <= 256 rows are live,
 *                       if larger than sync_every
 *   "obj_inblk_max" 768 objective launches of <= this many blocks search their grid
 *                       cell inside the block (0 = never)
 *   "obj_sort"       1  objective jobs evaluated in grid-cell order
 *   "nn_pipe"        1  rvs_template_nn(_arms): the wide last layer through the kernel
 *                       whose epilogue runs under the next tile's products (K of 128,
 *                       200 or 256 inputs); 0 = the generic kernel (same bits)
 * Both return 0, or RVS_E_ARG for an unknown name / NULL value pointer.
 * ---------------------------------------------------------------------- */
int rvs_option_set(const char *name, int value);
int rvs_option_get(const char *name, int *value);

/* ------------------------------------------------------------------------
 * A3  polylinear template evaluation on a regular n-D grid
 *     replaces spec_inter.GridInterp.__call__ (spec_inter.py:134-194),
 *     GridOutsideCheck.__call__ (:77-92) and LogParamMapper.forward
 *     (read_grid.py:127-145), composed as SpecInterpolator.eval/outsideFlag
 *     (spec_inter.py:257-286) + the MAX_VAL guard of getCurTempl
 *     (spec_fit.py:392-397).
 *
 * dats      float32 [ngrid, ntp]  log-flux rows (interpdat_%s.npy)
 * idgrid    int64   [prod(lens)]  row of dats or -1 (C order)
 * uvecs     float64 [sum(lens)]   concatenated unique mapped grid values
 * lens      int32   [ndim] (host)
 * vecs_s    float64 [ngrid, ndim] grid points divided by ptp (KD-tree space)
 * ptp       float64 [ndim] (host) np.ptp(vec, axis=1): the query is p / ptp
 * log_mask  bit i set -> parameter i is mapped through log10
 * params    float64 [B, ndim]     physical parameters
 * templ     float64 [B, ntp]  out  exp'ed template
 * outside   float64 [B]       out  0 inside; KD distance outside; NaN if the
 *                                   template is non finite / > 1e100 outside
 * cellinfo  int32   [B, 2+2^ndim] out (nullable) {mode, nearest, vertex ids}
 *           mode 0 = polylinear, 1 = nearest neighbour, 2 = non finite params
 * weights   float64 [B, 2^ndim]  out (nullable) polylinear weights
 * ---------------------------------------------------------------------- */
int rvs_template_polylinear(const float *dats, int64_t ngrid, int ntp,
                            const int64_t *idgrid, const double *uvecs,
                            const int32_t *lens, int ndim, const double *vecs_s,
                            const double *ptp, uint32_t log_mask,
                            int exp_flag, const double *params, int B,
                            double *templ, double *outside, int32_t *cellinfo,
                            double *weights, void *stream);

/* ------------------------------------------------------------------------
 * A3 on an irregular grid: Delaunay evaluator, replaces spec_inter.TriInterp
 * (spec_inter.py:11-59) for `interpolation_type = 'triangulation'` libraries
 * (make_nd without --regulargrid): find_simplex (exhaustive, scipy's inside
 * test with eps = 100*DBL_EPSILON, lowest simplex id) + barycentric blend of
 * ndim+1 rows + exp, and the same blend of `extraflags` as outside flag
 * (+ the MAX_VAL guard of getCurTempl).  No containing simplex -> NaN template,
 * NaN outside flag.
 * dats       float64 [npts, ntp]      log-flux rows incl. the padded edge points
 * simplices  int32   [nsimplex, ndim+1]
 * transform  float64 [nsimplex, ndim+1, ndim]  scipy Delaunay.transform
 * extraflags float64 [npts]
 * simplex    int32   [B]   out (required): simplex id or 0x7fffffff
 * weights    float64 [B, ndim+1] out (nullable)
 * ---------------------------------------------------------------------- */
int rvs_template_tri(const double *dats, int ntp, const int32_t *simplices,
                     const double *transform, const double *extraflags,
                     int nsimplex, int ndim, uint32_t log_mask, int exp_flag,
                     const double *params, int B, double *templ, double *outside,
                     int32_t *simplex, double *weights, void *stream);

/* ... with find_simplex through a bucket grid instead of the exhaustive search (the
 * same answer: the lowest simplex id that passes scipy's inside test,
 * spec_inter.py:11-59 / Delaunay.find_simplex): cell c of a uniform grid over the
 * mapped parameter space lists, ascending, every simplex whose bounding box (grown by
 * 1e-9 of the grid's extent) overlaps it.  The cell of a mapped coordinate x in
 * dimension d is clamp(floor((x - lo[d]) * inv_w[d]), 0, n[d] - 1), cells in C order
 * over the dimensions.  List number ncell (behind the last cell) holds the simplices
 * whose boxes overlap too many cells to be entered in each: every query tests it as
 * well, the lower id of the two finds wins.  The struct is read on the host;
 * cell_start [ncell + 2] and cell_list [cell_start[ncell + 1]] are device arrays. */
typedef struct rvs_tri_buckets {
  const int32_t *cell_start, *cell_list;
  double lo[6], inv_w[6];
  int32_t n[6];
} rvs_tri_buckets;
int rvs_template_tri_buckets(const double *dats, int ntp, const int32_t *simplices,
                             const double *transform, const double *extraflags,
                             int nsimplex, int ndim, uint32_t log_mask,
                             int exp_flag, const rvs_tri_buckets *buckets,
                             const double *params, int B, double *templ,
                             double *outside, int32_t *simplex, double *weights,
                             void *stream);

/* ------------------------------------------------------------------------
 * A6  rotational broadening; replaces spec_fit.convolve_vsini /
 *     compute_vsini_kernel (spec_fit.py:495-682).  vsini[b] <= 0, NaN or
 *     R < 1e-9 copies the row (as does a non finite `outside[b]`, nullable,
 *     spec_fit.py:398-404).  lnstep = log(lam[1]/lam[0]) of the log-uniform
 *     template grid.  In place (out == templ) is NOT allowed.
 * ---------------------------------------------------------------------- */
int rvs_vsini_convolve(const double *templ, const double *vsini,
                       const double *outside, double lnstep, double eps,
                       int ntp, int B, double *out, void *stream);

/* ------------------------------------------------------------------------
 * A7  natural cubic spline through (knots, ys[b]); replaces `construct`
 *     (src/spliner.c:7-60).  form 0: coef[b, i, 0..3] = A,B,C,D of interval i
 *     exactly as the reference (i < ntp-1; row ntp-1 is zero padding -- in
 *     form 1 it holds {y_last, 0, 0, 0}; h is implied by knots).  form 1: the SAME cubic in powers of dl = x - x_i,
 *     {y_i, b, c, d} with S = y + dl (b + dl (c + dl d)) -- 3 fma per evaluation,
 *     the form the fused chi^2 kernels consume.  form | 2: the caller asserts
 *     that neighbouring knot spacings agree to ~1 % (uniform or log-uniform
 *     grids, what `evaler` requires anyway): both Thomas recurrences then
 *     contract by ~0.27 per row and are evaluated in independent 32-row-overlap
 *     windows (error < 1e-18 relative) instead of by exact chunk carries; this
 *     needs `factors` from rvs_spline_factors (pivots, h, 1/h of the knot grid,
 *     computed once per template grid: 5 arrays of ntp doubles, followed -- for ntp <=
 *     8192 -- by 1/h_u, 1/h_{u+1}, g_u, e_u (records of four) and c_u (pairs) once
 *     more in the order the fused objective kernel's 512 threads own their rows,
 *     5 x 8192 doubles;
 *     rvs_spline_factors_len(ntp) in all); NULL otherwise.
 * ---------------------------------------------------------------------- */
int64_t rvs_spline_factors_len(int ntp);   /* doubles `factors` must hold */
int rvs_spline_factors(const double *knots, int ntp, double *factors,
                       void *stream);
int rvs_spline_construct(const double *knots, const double *ys, int ntp, int B,
                         int form, const double *factors, double *coef,
                         void *stream);

/* A7  replaces `evaler` (src/spliner.c:71-108) for B splines sharing the
 * knots: ret[b, i] = S_b(evalx[b, i]); pos (nullable) receives the integer
 * interval index (int)((log x - log x0)/logstep) computed exactly as the
 * reference does; status[b] gets RVS_ST_SPLINE_RANGE / _GRID. */
int rvs_spline_eval(const double *knots, const double *coef, int ntp,
                    int log_step, const double *evalx, int neval, int B,
                    double *ret, int32_t *pos, int32_t *status, void *stream);

/* ------------------------------------------------------------------------
 * A7-eval + A10 + A11 fused: Doppler resample of a spline template onto the
 * observed pixels and continuum-marginalised -2 log L on a velocity grid;
 * replaces the vel loop of spec_fit.find_best over spec_fit.get_chisq
 * (spec_fit.py:797-989, 1060-1071) for ONE spectral arm.
 *
 * Two calls: rvs_chisq_prepare once per (arm, batch of spectra) builds the
 * velocity-independent per-pixel terms into `work` (rvs_chisq_work_size
 * doubles); rvs_chisq_grid may then be called many times (first guess,
 * refinement rounds, optimiser steps) on the same `work`.
 *
 * lam     [npix]        observed wavelengths of the arm (shared by the batch)
 * polysT  [npix, npoly] continuum basis, pixel-major (get_poly_basis^T)
 * spec, espec [S, npix]
 * knots   [ntp], coef [Tn, ntp, 4]   from rvs_spline_construct(form = 1)
 * knots_host3  HOST pointer to the first three knots: the uniformity test of
 *          spliner.c:84-96 is done on the host; returns -3 where evaler
 *          returns -2
 * job_spec, job_templ int32 [J] (nullable = identity): job j fits spectrum
 *          job_spec[j] with template job_templ[j]
 * vels    float64, job j uses vels + j*vel_stride (vel_stride 0 = shared grid)
 * espec_sys  systematic error added in quadrature (spec_fit.py:933-940)
 * penalty [J] (nullable) added to every velocity of job j (outside*badchi,
 *          spec_fit.py:895-896); a non finite penalty means "template not
 *          usable": the arm contributes 1000*badchi (spec_fit.py:888-893).
 * beta    out = beta*out + value  (0 first arm, 1 following arms)
 * pack_min_jobs  the Nv % 64 left-over velocities of a job (16 of the 400-point
 *          grid) share a wave with those of other jobs when J >= this
 *          (0 = library default 2000, 1 = always, < 0 = never); where a
 *          velocity is computed does not change its value.  The packed launch
 *          runs on a library-owned side stream (one per host thread and
 *          device), forked from and joined to `stream` inside the call: to the
 *          caller the call is ordered on `stream` like any other.
 * out     [J, Nv];   status int32 [J] OR-ed (caller zeroes it)
 * ---------------------------------------------------------------------- */
int64_t rvs_chisq_work_size(int npix, int S);
int rvs_chisq_prepare(const double *lam, const double *spec,
                      const double *espec, int npix, int S,
                      const double *knots_host3, int log_step, double espec_sys,
                      double *work, void *stream);
int rvs_chisq_grid(const double *lam, const double *polysT, const double *work,
                   int npix, int npoly, int S, const double *knots,
                   const double *coef, int ntp, int Tn, int log_step,
                   const int32_t *job_spec, const int32_t *job_templ, int J,
                   const double *vels, int64_t vel_stride, int Nv,
                   const double *penalty, double badchi, double beta,
                   int pack_min_jobs, double *out, int32_t *status,
                   void *stream);

/* Grid sets (ABI 7): the S spectra of an arm observed on G different wavelength
 * grids (SDSS-style objects: every spectrum has its own `lam`, spec_fit.py:70-145,
 * tests/test_sdss.py).  lam [G, npix]: grid g in row g, a grid with fewer than
 * npix pixels padded by repeating its last wavelength; grid_id int32 [S]: the grid
 * of spectrum s; on the padding the spectra carry espec = +inf (weight 0, no term
 * in sum log e -- read as the marker ONLY when G > 1: on a single grid an infinite
 * error is data and gives the reference's log(inf), a non-finite likelihood the
 * status word reports) and the basis rows are 0; polysT of grid g starts at
 * polysT + g * polys_stride.  work: rvs_chisq_work_size_g(npix, S, G) doubles from
 * rvs_chisq_prepare_g.  G = 1 (grid_id NULL) is the shared-grid call.  With G > 1
 * the left-over velocities of a job are not packed with those of other jobs.
 * pen_scale [S] or NULL: per-spectrum factor on badchi (see rvs_point_arm). */
int64_t rvs_chisq_work_size_g(int npix, int S, int G);
int rvs_chisq_prepare_g(const double *lam, const double *spec,
                        const double *espec, int npix, int S, int G,
                        const double *knots_host3, int log_step, double espec_sys,
                        double *work, void *stream);
int rvs_chisq_grid_g(const double *lam, const double *polysT, const double *work,
                     int npix, int npoly, int S, const int32_t *grid_id, int G,
                     int64_t polys_stride, const double *knots,
                     const double *coef, int ntp, int Tn, int log_step,
                     const int32_t *job_spec, const int32_t *job_templ, int J,
                     const double *vels, int64_t vel_stride, int Nv,
                     const double *penalty, double badchi, double beta,
                     int pack_min_jobs, const double *pen_scale, double *out,
                     int32_t *status, void *stream);

/* A9  the same with a banded resolution matrix applied to the resampled
 * template before the fit: replaces convolve_resol / ResolMatrix
 * (spec_fit.py:54-67, 474-492) as used by get_chisq (:920-929).
 * taps [S or 1, npix, nd] row-major: taps[s, k, d] = R_s[k, k - (nd-1)/2 + d]
 * (0 outside the matrix), nd odd <= 33; taps_stride = npix*nd, or 0 when every
 * spectrum shares one matrix (the `resol_params` case). */
int rvs_chisq_grid_resol(const double *lam, const double *polysT,
                         const double *work, int npix, int npoly, int S,
                         const double *knots, const double *coef, int ntp,
                         int Tn, int log_step, const double *taps, int nd,
                         int64_t taps_stride, const int32_t *job_spec,
                         const int32_t *job_templ, int J, const double *vels,
                         int64_t vel_stride, int Nv, const double *penalty,
                         double badchi, double beta, double *out,
                         int32_t *status, void *stream);
/* ... for spectra of an arm on G wavelength grids of their own (grid sets, above:
 * lam [G, npix], work from rvs_chisq_prepare_g, polysT of grid g at polysT + g *
 * polys_stride), each with its own resolution matrix (spec_fit.py:922-929 accepts
 * any SpecData.resolution; tests/test_sdss.py): taps [S, npix, nd] with npix the
 * longest grid -- the rows behind a spectrum's own pixels, and every tap that
 * refers to a pixel behind them, are zero.  pen_scale as in rvs_chisq_grid_g.
 * G = 1 (grid_id NULL) is rvs_chisq_grid_resol. */
int rvs_chisq_grid_resol_g(const double *lam, const double *polysT,
                           const double *work, int npix, int npoly, int S,
                           const int32_t *grid_id, int G, int64_t polys_stride,
                           const double *knots, const double *coef, int ntp,
                           int Tn, int log_step, const double *taps, int nd,
                           int64_t taps_stride, const int32_t *job_spec,
                           const int32_t *job_templ, int J, const double *vels,
                           int64_t vel_stride, int Nv, const double *penalty,
                           double badchi, double beta, const double *pen_scale,
                           double *out, int32_t *status, void *stream);

/* ------------------------------------------------------------------------
 * get_chisq(full_output=True) for one velocity per job and one arm
 * (spec_fit.py:941-961), and get_chisq_continuum (spec_fit.py:739-783) when
 * unit_template == 1 (template == 1, knots/coef ignored); unit_template == 2:
 * get_chisq0 itself (spec_fit.py:306-354) -- the template is given ON THE PIXELS,
 * row job_templ[j] of `coef` read as [Tn, npix] doubles (knots, ntp, vel, cform
 * ignored; espec of ones = "already divided by the uncertainty").  cform = the form of
 * the spline records (see rvs_spline_construct).  fast_interp: the template
 * value is the nearest knot at or above x instead of the spline
 * (spec_fit.py:913-918; needs cform = 1).  taps (nullable): resolution
 * matrix rows as in rvs_chisq_grid_resol, applied to the template (or to 1).
 * coeffs [J, npoly], model/raw_model [J, npix] (nullable), chisq [J] (-2logL
 * of the arm), true_chisq [J] over pixels with badmask==0, ngood int32 [J].
 * ---------------------------------------------------------------------- */
int rvs_chisq_full(const double *lam, const double *polysT, const double *spec,
                   const double *espec, const uint8_t *badmask, int npix,
                   int npoly, int S, const double *knots, const double *coef,
                   int ntp, int Tn, int log_step, int cform, int unit_template,
                   const int32_t *job_spec, const int32_t *job_templ, int J,
                   const double *vel, double espec_sys, int fast_interp,
                   const double *taps, int nd, int64_t taps_stride,
                   double *chisq, double *coeffs,
                   double *model, double *raw_model, double *true_chisq,
                   int32_t *ngood, int32_t *status, void *stream);
/* ... on grid sets (see rvs_chisq_grid_g): lam [G, npix], polysT per grid; the
 * padding of a short grid does not count in ngood / true_chisq, model is 0 there. */
int rvs_chisq_full_g(const double *lam, const double *polysT, const double *spec,
                     const double *espec, const uint8_t *badmask, int npix,
                     int npoly, int S, const double *knots, const double *coef,
                     int ntp, int Tn, int log_step, int cform, int unit_template,
                     const int32_t *job_spec, const int32_t *job_templ, int J,
                     const double *vel, double espec_sys, int fast_interp,
                     const double *taps, int nd, int64_t taps_stride,
                     double *chisq, double *coeffs, double *model,
                     double *raw_model, double *true_chisq, int32_t *ngood,
                     int32_t *status, const int32_t *grid_id, int G,
                     int64_t polys_stride, void *stream);

/* ------------------------------------------------------------------------
 * A13  get_chisq_continuum (spec_fit.py:739-783) for a whole batch of one arm:
 * continuum-only fit (template == 1), true chi^2 over pixels with badmask == 0
 * and their count.  One wave per spectrum, lanes = pixels; a Cholesky failure (numerically
 * singular basis) is flagged and gives NaN -- rvs_chisq_full(unit_template=1)
 * is the slower entry point with the eigen (SVD) fallback.
 * polysT [npix, npoly]; spec, espec [S, npix]; badmask uint8 [S, npix] (nullable)
 * unit_templ [S, npix] (nullable): R_s @ 1 when the spectra carry a resolution
 * matrix (spec_fit.py:765-767)
 * work: unused since round 2 (nullable; rvs_chisq_continuum_work_size returns 8)
 * chisq [S] (-2 log L, nullable), true_chisq [S], ngood int32 [S]
 * ---------------------------------------------------------------------- */
int64_t rvs_chisq_continuum_work_size(int npoly, int S);
int rvs_chisq_continuum(const double *polysT, const double *spec,
                        const double *espec, const uint8_t *badmask,
                        const double *unit_templ, int npix, int npoly, int S,
                        void *work, double *chisq,
                        double *true_chisq, int32_t *ngood, int32_t *status,
                        void *stream);
int rvs_chisq_continuum_g(const double *polysT, const double *spec,
                          const double *espec, const uint8_t *badmask,
                          const double *unit_templ, int npix, int npoly, int S,
                          void *work, double *chisq, double *true_chisq,
                          int32_t *ngood, int32_t *status,
                          const int32_t *grid_id, int G, int64_t polys_stride,
                          void *stream);

/* ------------------------------------------------------------------------
 * A11 at one velocity per job: the objective of the optimiser stage,
 * chisq_func0 (vel_fit.py:205-226) = get_chisq (spec_fit.py:797-989) for J
 * (spectrum, template, velocity) triples, ALL arms of the spectrum in one
 * launch.  One 256-thread block per (job, arm), threads = pixels; the residual
 * norm ||D - a.ST||^2 is formed explicitly (spec_fit.py:249) so the value can
 * be finite-differenced (Hessian, vel_fit.py:699-725).
 * Per arm (plain struct of device pointers, passed by value from the host):
 *   lam [npix], polysT [npix, npoly] (plain get_basis, NOT orthonormalised),
 *   spec, espec [S, npix], work = the rvs_chisq_prepare buffer of the arm,
 *   knots [ntp], coef [Tn, ntp, 4] form-1 records, penalty [J] (nullable;
 *   a NaN/inf entry means "template unusable": + 1000*badchi, arm skipped).
 * scratch: rvs_chisq_point_work_size(J, narm) bytes.
 * out[j] = sum over arms of (chisq + penalty).
 * ---------------------------------------------------------------------- */
#define RVS_MAX_ARMS 4
typedef struct rvs_point_arm {
  const double *lam, *polysT, *spec, *espec, *work, *knots, *coef, *penalty;
  const double *taps;   /* A9 resolution matrix rows [S or 1, npix, nd] or NULL */
  int64_t taps_stride;  /* npix*nd, or 0 when all spectra share one matrix */
  double espec_sys;     /* systematic error in quadrature (spec_fit.py:933-940);
                           `work` must have been prepared with the same value */
  int32_t npix, S, ntp, log_step, nd;
  int32_t fast_interp;  /* nearest-knot template instead of the spline (:913-918) */
  /* wavelength grids of the arm (ABI 7): G <= 1 -- one grid, lam [npix], polysT
   * [npix, npoly], work from rvs_chisq_prepare; G > 1 -- spectrum s is observed on
   * grid grid_id[s]: lam [G, npix], polysT of grid g at polysT + g*polys_stride,
   * work from rvs_chisq_prepare_g (a grid shorter than npix is padded: lam repeats
   * its last value, espec = +inf, basis rows 0).  No resolution matrix with G > 1. */
  const int32_t *grid_id;
  int64_t polys_stride;
  int32_t G, reserved_;
  /* per-spectrum factor on `badchi` [S] or NULL (read from arm 0): badchi is
   * 10 x the pixel count of the SPECTRUM (spec_fit.py:863), which differs between
   * the spectra of a grid set -- the scalar argument carries the largest, this the
   * ratio */
  const double *pen_scale;
} rvs_point_arm;
int64_t rvs_chisq_point_work_size(int J, int narm);
int rvs_chisq_point(const rvs_point_arm *arms, int narm, int npoly,
                    const int32_t *job_spec, const int32_t *job_templ, int J,
                    const double *vel, double badchi, void *scratch,
                    double *out, int32_t *status, void *stream);

/* ------------------------------------------------------------------------
 * The same objective as ONE kernel per evaluation for regular-grid libraries:
 * polylinear gather + exp (A3/A5), rotational FIR (A6), windowed natural-spline
 * solve (A7) and the chi^2 of rvs_chisq_point (A10/A11), one 512-thread block
 * per (job, arm) with the whole template in LDS (3*ntp doubles <= 154 KB); no
 * spline record goes to HBM.  Replaces, for the optimiser's ~850 calls per
 * spectrum, the chain rvs_template_polylinear -> rvs_vsini_convolve ->
 * rvs_spline_construct -> rvs_chisq_point (same arithmetic per phase).
 * Per arm: `pt` as in rvs_chisq_point (coef, penalty, taps unused; no
 * fast_interp), the polylinear library as in rvs_template_polylinear, the
 * rvs_spline_factors of the template grid, ln-step of the grid for the
 * rotational kernel.  params [J, ndim], vsini [J] (nullable = no rotation),
 * vel [J]; scratch: rvs_objective_work_size(J, narm) bytes;
 * out[j] = sum over arms of chisq + outside*badchi (NaN outside: + 1000*badchi).
 * outside_penalty: bit 0 = add the outside*badchi penalty; bit 1
 * (RVS_OBJ_STATUS_STORE) = status[j] is overwritten instead of OR-ed into (the
 * optimiser's per-call scratch needs no clearing launch).
 * RVS_E_ARG: ntp < 32 or > rvs_objective_max_ntp(npoly); and, from npoly = 11 on
 * (the waves' partial sums live in the template's LDS), an arm with
 * 2*npix > ntp or 2*ntp < 8*(npoly*(npoly+3)/2 + 1) -- use the chain of
 * stand-alone kernels there.
 * ---------------------------------------------------------------------- */
#define RVS_OBJ_STATUS_STORE 2
/* bit 2: the per-arm results stay in `scratch` ([narm, J] chi^2, [narm, J] outside,
 * [narm, J] int32 status), out / status are not written: rvs_nm_run folds the sum
 * over the arms into its own bookkeeping kernels */
#define RVS_OBJ_NO_SUM 4
typedef struct rvs_objective_arm {
  rvs_point_arm pt;
  const float *dats;
  const int64_t *idgrid;
  const double *uvecs, *vecs_s, *factors;
  int64_t ngrid;
  double lnstep;
  double ptp[6];
  int32_t lens[6];
  int32_t ntp, ndim;
  uint32_t log_mask;
  int32_t exp_flag;
} rvs_objective_arm;
int rvs_objective_max_ntp(int npoly); /* largest template grid that fits LDS */
int64_t rvs_objective_work_size(int J, int narm);
int rvs_objective_fused(const rvs_objective_arm *arms, int narm, int npoly,
                        const double *params, const double *vsini,
                        const int32_t *job_spec, int J, const double *vel,
                        double badchi, int outside_penalty, void *scratch,
                        double *out, int32_t *status, void *stream);

/* The same objective for evaluators that are not a grid gather (the MLP of
 * rvs_template_nn): the unbroadened template of job j on arm a is row j of
 * templ[a] ([J, ntp] float64, device), its outside flag outside[a][j]
 * (SpecInterpolator.outsideFlag, spec_inter.py:257-272; the MAX_VAL guard of
 * getCurTempl, spec_fit.py:392-397, is applied here).  `templ` / `outside` are
 * HOST arrays of narm device pointers.  Of rvs_objective_arm only pt, factors,
 * lnstep and ntp are read.  Rotational broadening, spline solve and chi^2 are
 * the code of rvs_objective_fused: equal arithmetic. */
int rvs_objective_from_template(const rvs_objective_arm *arms, int narm,
                                int npoly, const double *const *templ,
                                const double *const *outside,
                                const double *vsini, const int32_t *job_spec,
                                int J, const double *vel, double badchi,
                                int outside_penalty, void *scratch, double *out,
                                int32_t *status, void *stream);

/* Both with a job count that lives on the device (ABI 8): only the first
 * min(J, njobs_dev[0]) jobs are evaluated, out[] / status[] of the others are left
 * as they are; J still sizes the launch, the scratch and every array.  The
 * lock-step optimiser (rvs_nm_run) knows on the host only an upper bound of the
 * simplices that are still running and of those that need the second point of a
 * round; the exact counts are its device counters.  NULL = all J. */
int rvs_objective_fused_n(const rvs_objective_arm *arms, int narm, int npoly,
                          const double *params, const double *vsini,
                          const int32_t *job_spec, int J,
                          const int32_t *njobs_dev, const double *vel,
                          double badchi, int outside_penalty, void *scratch,
                          double *out, int32_t *status, void *stream);
int rvs_objective_from_template_n(const rvs_objective_arm *arms, int narm,
                                  int npoly, const double *const *templ,
                                  const double *const *outside,
                                  const double *vsini, const int32_t *job_spec,
                                  int J, const int32_t *njobs_dev,
                                  const double *vel, double badchi,
                                  int outside_penalty, void *scratch, double *out,
                                  int32_t *status, void *stream);

/* ------------------------------------------------------------------------
 * Host side of the second minimiser of vel_fit.process (vel_fit.py:653-658:
 * scipy.optimize.minimize(method='BFGS', options={'hess_inv0': ...})): S
 * independent BFGS runs (scipy's _minimize_bfgs with its wolfe1 / wolfe2 line
 * searches and 2-point forward-difference gradients) advanced in lock-step.
 * The caller owns the objective:
 *     h = rvs_bfgs_begin(S, n, x0 [S,n], hess_inv0 [n,n] or NULL, gtol, c1, c2,
 *                        xrtol, maxiter (<= 0: 200 n))
 *     while ((rows = rvs_bfgs_pending(h, idx, X, cap)) > 0) {
 *         F[r] = objective(spectrum idx[r], point X[r, :]),  r < rows
 *         rvs_bfgs_feed(h, F, rows);
 *     }
 *     rvs_bfgs_result(h, x [S,n], fun [S], nit, nfev, status [S] (scipy's
 *                     warnflag), hess_inv [S,n,n] or NULL, &rounds);
 *     rvs_bfgs_end(h);
 * cap >= S (n + 1) rows always suffices.  n <= 16.  No GPU work in here.
 * ---------------------------------------------------------------------- */
void *rvs_bfgs_begin(int S, int n, const double *x0, const double *hess_inv0,
                     double gtol, double c1, double c2, double xrtol,
                     int maxiter);
int64_t rvs_bfgs_pending(void *h, int64_t *idx, double *X, int64_t cap_rows);
int rvs_bfgs_feed(void *h, const double *F, int64_t nrows);
int rvs_bfgs_result(void *h, double *x, double *fun, int32_t *nit,
                    int32_t *nfev, int32_t *status, double *hess_inv,
                    int64_t *rounds);
void rvs_bfgs_end(void *h);

/* ------------------------------------------------------------------------
 * The tables that depend only on a wavelength grid, for G grids at once (one per
 * arm on the fast path; one per SPECTRUM for SDSS-style objects): built on the
 * device because on the host they cost ~10 ms per grid (LAPACK QR, searches).
 *
 * rvs_basis_build: the continuum basis get_basis (spec_fit.py:148-176) -- rbf != 0:
 * 1, x, x^2 and npoly-3 Gaussians at `cen` (device, np.linspace(-1, 1, npoly-3));
 * rbf == 0: Chebyshev T_i by numpy's Clenshaw recursion -- on x = the grid's own
 * range mapped to [-1, 1]; raw [G, npix+1, npoly] pixel-major (rows from a grid's
 * pixel count on are 0); ortho (nullable) the same function space orthonormalised
 * over the grid's pixels (modified Gram-Schmidt, twice) with logdet[g] =
 * 2 sum log R_jj, what the velocity-grid kernel reads (see rvs_chisq_grid).
 * lam [G, npix]; npix_g int32 [G] or NULL (every grid has npix pixels).
 * Limits: npoly <= 32 (what rvs_chisq_full takes), npix <= 16384; RVS_E_ARG beyond. */
int rvs_basis_build(const double *lam, const int32_t *npix_g, int G, int npix,
                    int npoly, int rbf, const double *cen, double *raw,
                    double *ortho, double *logdet, void *stream);
/* rvs_ccf_tables_build: the grid-dependent tables of rvs_ccf_preprocess(_g):
 * xind / rw [G, nfft] (make_ccf.py:355-357, 394-399; ccf_lam [nfft] = the FFT grid's
 * wavelengths, device), and with `continuum` the B-spline form of the k = 2
 * interpolating spline through the continuum nodes (make_ccf.py:155-164; FITPACK
 * knots and fpbspl): Eb [G, npix, 3], El [G, npix], istart [G, nnode], bin_start
 * [G, nnode + 1] (make_ccf.py:128-143, scipy.stats.binned_statistic ranges) from
 * nodes [G, nnode] / edges [G, nnode + 1] / nnode_g [G] (make_ccf.py:123-131,
 * computed by the caller: a log and an exp per node).  The collocation matrix and
 * its inverse (Cinv) stay with the caller (nnode <= 24). */
int rvs_ccf_tables_build(const double *lam, const int32_t *npix_g, int G, int npix,
                         const double *ccf_lam, int nfft, int continuum,
                         const double *nodes, const double *edges,
                         const int32_t *nnode_g, int nnode, int32_t *xind,
                         double *rw, double *Eb, int32_t *El, int32_t *istart,
                         int32_t *bin_start, void *stream);

/* ------------------------------------------------------------------------
 * A12  grid summary; replaces the tail of spec_fit.find_best
 * (spec_fit.py:1072-1092) and _quadratic_interp_min (:992-1015).
 * chisq [G, Np, Nv] (velocity fastest); vels + g*vel_stride -> [Nv];
 * nvel (nullable int32 [G]) = number of valid velocities of group g.
 * res [G, 8] = best_chi, best_vel, vel_err, kurtosis, skewness, i1 (vel idx),
 *              i2 (template idx), spare;   probs [G, Nv] (nullable)
 * ---------------------------------------------------------------------- */
int rvs_grid_moments(const double *chisq, const double *vels,
                     int64_t vel_stride, const int32_t *nvel, int G, int Np,
                     int Nv, int quadratic, double *res, double *probs,
                     int32_t *status, void *stream);

/* ------------------------------------------------------------------------
 * A15  CCF pre-processing of one arm; replaces make_ccf.preprocess_data
 * (make_ccf.py:330-414) with interp_masker (:288-327), get_continuum
 * (:105-152) and fit_resid (:155-164).  The k=2 interpolating continuum
 * spline is linear in its node values p: with C the B-spline collocation
 * matrix at the nodes, S(lam_k) = sum_{q<3} Eb[k,q] * (C^-1 p)[El[k]+q].  The
 * robust soft-L1 fit is a device Levenberg-Marquardt on the same objective.
 *
 * Eb float64 [npix, 3], El int32 [npix]   B-spline basis of every pixel
 * Cinv float64 [2, nnode, nnode]          inverse collocation matrix C^-1, then C
 * istart int32 [nnode-1]                  first pixel of every knot interval
 * bin_start int32 [nnode+1]  pixel ranges of the binned-median start
 * xind int32 [nfft], rw float64 [nfft]   rebin tables (xind<0: no coverage)
 * outputs: proc_spec, proc_ivar [B, nfft]; sse [B] = sum proc_spec^2 proc_ivar;
 *          cont [B, npix] (nullable), pfit [B, nnode] (nullable)
 * ---------------------------------------------------------------------- */
int rvs_ccf_preprocess(const double *lam, const double *spec,
                       const double *espec, const uint8_t *badmask, int npix,
                       int B, int continuum, const double *Eb, const int32_t *El,
                       const double *Cinv, const int32_t *istart, int nnode,
                       const int32_t *bin_start, const int32_t *xind,
                       const double *rw, int nfft, double maxerr,
                       double *proc_spec, double *proc_ivar, double *sse,
                       double *cont, double *pfit, int32_t *status,
                       void *stream);
/* ... on grid sets: spectrum b is observed on grid grid_id[b] with npix_g[g]
 * pixels and nnode_g[g] continuum nodes (make_ccf.py:123-131: the node count
 * follows the grid's wavelength range).  Every table holds one slice per grid:
 * lam [G, npix], Eb [G, npix, 3], El [G, npix], Cinv [G, 2, nnode, nnode] (grid g's
 * two nnode_g x nnode_g matrices packed at the start of its slice), istart
 * [G, nnode], bin_start [G, nnode + 1], xind / rw [G, nfft]; npix / nnode = the
 * largest grid = the row strides of spec, espec, badmask, cont / pfit. */
int rvs_ccf_preprocess_g(const double *lam, const double *spec, const double *espec,
                         const uint8_t *badmask, int npix, int B, int continuum,
                         const double *Eb, const int32_t *El, const double *Cinv,
                         const int32_t *istart, int nnode,
                         const int32_t *bin_start, const int32_t *xind,
                         const double *rw, int nfft, double maxerr,
                         double *proc_spec, double *proc_ivar, double *sse,
                         double *cont, double *pfit, int32_t *status,
                         const int32_t *grid_id, const int32_t *npix_g,
                         const int32_t *nnode_g, void *stream);

/* ------------------------------------------------------------------------
 * A14  FFT cross-correlation of one arm against T templates; replaces the
 * per-arm body of fitter_ccf.fit (fitter_ccf.py:112-161, 189-216).
 *
 * tfft, tfft2 complex128 [T, nfft/2+1] (ccfdat_%s.npz 'fft','fft2')
 * twid complex128 [nfft/2]   exp(+2 pi i k / nfft)
 * lag_pos int32 [nlag]  position of lag n = subind[l] in the LDS image of the
 *         half-size inverse FFT viewed as doubles: with p = rvs_ccf_fft_pos(nfft,
 *         n>>1) (digit-reversed output order of the radix-8 DIF passes),
 *         lag_pos = 2*p + (n&1)   (the image is not padded since round 2)
 * lag_vel float64 [nlag] ascending lag velocities (fitter_ccf.py:136-154)
 * ilo int32 [nvel], vgrid float64 [nvel]        linear-interp tables
 * chisq [B, T, nvel]: out = beta*out + interp(-2 c0 + c1) (continuum) or
 *                      interp(-c0^2/c1)
 * prune uint8 [n2/64 + n2/8], n2 = nfft/2 (nullable; used when n2 is a power of
 *         8): only the lags in lag_pos are read back, so the last two radix-8
 *         passes compute selected outputs only.  With p as above,
 *         prune[n2/64 + (p>>3)] has bit (p&7) set for every needed p, and
 *         prune[p>>6] has bit ((p>>3)&7) set (the outputs of the second-last
 *         pass that feed a needed 8-group).  Values are bit-identical to the
 *         unpruned transform.
 * work  complex128 [B, 2, nfft/2+1] scratch for conj rfft of spec*ivar, ivar
 * Kernels behind the call (same interface, results equal to a few ulp): at nfft
 * 8192 (with `prune`) and 4096, nlag and nvel <= 512, T >= 2, one persistent
 * wave-specialised block per spectrum walks the T templates (ccf_xcorr_ws_kernel /
 * _ws2_kernel, csrc/ccf_fft.hip; without continuum normalisation it takes one of the
 * two correlations per iteration); otherwise one block per (spectrum, template).
 * rvs_option_set("xc_ws", 0) forces the latter.
 * ---------------------------------------------------------------------- */
int rvs_ccf_fft_pos(int nfft, int f);  /* host helper, see lag_pos */
int rvs_ccf_xcorr(const double *proc_spec, const double *proc_ivar, int nfft,
                  int B, const double *tfft, const double *tfft2, int T,
                  const double *twid, int continuum, const int32_t *lag_pos,
                  const double *lag_vel, int nlag, const int32_t *ilo,
                  const double *vgrid, int nvel, double beta,
                  const uint8_t *prune, double *chisq, double *work,
                  void *stream);

/* argmin over (template, velocity) + 3-point parabola (fitter_ccf.py:218-236).
 * sse [B] is added to every entry (total_sse).  res [B,4] = best_id, best_vel,
 * best_pix, min value; best_ccf [B, nvel]. */
int rvs_ccf_select(const double *chisq, const double *sse, int narm_sse, int B,
                   int T, const double *vgrid, int nvel, double *res,
                   double *best_ccf, int32_t *status, void *stream);

/* ------------------------------------------------------------------------
 * A4  NN template evaluator; replaces NNInterpolator.forward
 * (nn/NNInterpolator.py:14-91), Mapper.forward (:159-171) and
 * RVSInterpolator.__call__ (nn/RVSInterpolator.py:36-42): float32 MLP
 * (Linear+SiLU)*(nlayer-1), Linear, on f32-input MFMA, float64
 * exp(clip(.,-300,300)) epilogue.
 * W[l] float32 [dout_l, din_l] row-major (torch Linear.weight), b[l] [dout_l]
 * ---------------------------------------------------------------------- */
int rvs_template_nn(const double *params, int B, int ndim, uint32_t log_mask,
                    const double *M, const double *S, int nlayer,
                    const float *const *W, const float *const *b,
                    const int32_t *dims, float *act0, float *act1,
                    double *templ, void *stream);

/* Outside flag of an NN library; replaces OutsideInterpolator.__call__
 * (nn/RVSInterpolator.py:63-71) as SpecInterpolator.outsideFlag calls it on the
 * Mapper-transformed point (spec_inter.py:257-272, nn/NNInterpolator.py:159-171):
 * outside[j] = max(max_f xeqs[f].(p0,p1,1), max_f yeqs[f].(p2..,1), 0)^2.
 * xeqs [nfx, 3], yeqs [nfy, ndim-1]: scipy.spatial.ConvexHull(...).equations of
 * the first two / the remaining mapped coordinates of the training points
 * (built once by the caller); params, M, S, log_mask as rvs_template_nn;
 * mapped != 0: params ARE the Mapper's float64 output (M, S, log_mask unused).
 * A NaN parameter gives NaN (the arm is then skipped, spec_fit.py:888-893).
 * nfx == nfy == 0: a library without hulls (no outside check) -- outside[] is
 * zeroed on `stream`, nothing else is read. */
int rvs_nn_outside(const double *params, int B, int ndim, uint32_t log_mask,
                   const double *M, const double *S, int mapped,
                   const double *xeqs, int nfx, const double *yeqs, int nfy,
                   double *outside, void *stream);

/* ------------------------------------------------------------------------
 * the optimiser stage of vel_fit.process
 * (vel_fit.py:596-650).  S Nelder-Mead simplices (scipy's algorithm, which the
 * reference calls through scipy.optimize.minimize(method='Nelder-Mead',
 * options={fatol, xatol, initial_simplex, maxiter, maxfev=inf}), :627-637) live
 * in device memory and advance in lock-step; between the calls below the
 * caller evaluates the objective on (list, X) with the entry points above.
 *   sim [S, N+1, N], fsim [S, N+1] ordered ascending on entry; nit, nfev [S];
 *   flags [S]: bit0 active, bit1 converged, bit2 shrink pending;
 *   counts int32[8] (device): [0] |list1|, [1] |list2|, [2] |list3|,
 *   [3] simplices stepping this round, [4] simplices parked for a shrink,
 *   [5] rvs_nm_run: rows of a round that evaluates all four candidate points.
 * Every call takes `jbound`, a host-side upper bound of the job count (the
 * live count is read from `counts` on the device), so no host synchronisation
 * is needed inside a round; list/X entries in [count, jbound) are padded with
 * a copy of entry 0.
 * ---------------------------------------------------------------------- */
int rvs_nm_begin(int S, int N, double xatol, double fatol, int maxiter,
                 const double *sim, const double *fsim, const int32_t *nit,
                 int32_t *flags, int32_t *list1, double *X1, int32_t *counts,
                 int jbound, void *stream);
int rvs_nm_decide(int N, const double *sim, const double *fsim,
                  const int32_t *list1, const double *F1, int32_t *cases,
                  int32_t *pos2, int32_t *list2, double *X2, int32_t *counts,
                  int jbound, void *stream);
int rvs_nm_update(int N, double *sim, double *fsim, int32_t *nit, int32_t *nfev,
                  const int32_t *list1, const double *X1, const double *F1,
                  const int32_t *cases, const int32_t *pos2, const double *X2,
                  const double *F2, int32_t *flags, int32_t *counts, int jbound,
                  void *stream);
int rvs_nm_collect(int S, const int32_t *flags, int32_t *list3, int32_t *counts,
                   void *stream);
int rvs_nm_shrink_point(int N, int k, double *sim, const int32_t *list3,
                        double *X3, const int32_t *counts, int jbound,
                        void *stream);
int rvs_nm_shrink_store(int N, int k, double *sim, double *fsim, int32_t *nit,
                        int32_t *nfev, int32_t *flags, const int32_t *list3,
                        const double *F3, int32_t *counts, int jbound,
                        void *stream);

/* vel_fit.ParamMapper.forward + the range/finiteness guard of chisq_func +
 * VSiniMapper.to_vsini + the priors of chisq_func0 (vel_fit.py:95-254) for J
 * rows X [J, n] = (vel, [vsini], free stellar parameters).  src [ndim] (host):
 * column of X feeding stellar parameter i or -1 (fixed: fixed[S, ndim]);
 * vsini_col < 0: vsini_fixed[S]; vsini == NULL: no rotation.  Rows that
 * chisq_func answers with 1e30 get bad = 1 and harmless inputs (vel 0, safe[S,
 * ndim]).  rvs_proc_finish: F = bad ? 1e30 : chi + extra; job status bits of
 * the live rows (j < counts[cidx], or all when counts == NULL) are OR-ed into
 * spec_status[job_spec[j]]. */
int rvs_proc_map(int J, int n, int ndim, const double *X, const int32_t *list,
                 const int32_t *src, int vsini_col, const double *fixed,
                 const double *vsini_fixed, const double *safe,
                 const double *prior_mean, const double *prior_isig,
                 double min_vel, double max_vel, double max_vsini,
                 int32_t *job_spec, double *vel, double *vsini, double *params,
                 double *extra, int32_t *bad, void *stream);
int rvs_proc_finish(int J, const int32_t *counts, int cidx, const double *chi,
                    const double *extra, const int32_t *bad,
                    const int32_t *job_spec, const int32_t *job_status,
                    double *F, int32_t *spec_status, void *stream);

/* The rounds themselves, driven from C (what a caller of the entry points above
 * does per round, with the fused objective as chisq_func): blocks the calling
 * host thread until every simplex has stopped.  `m` holds the state arrays of
 * rvs_nm_* (fsim ordered, as for rvs_nm_begin), `o` the arguments of
 * rvs_proc_map / rvs_objective_fused / rvs_proc_finish and their row buffers
 * (capacity S rows; scratch: rvs_objective_work_size(S, narm) bytes).
 * stats (nullable) int64[3] = rounds, objective calls, rows evaluated. */
typedef struct rvs_nm_state {
  double *sim, *fsim, *X1, *X2, *F1, *F2;
  int32_t *nit, *nfev, *flags, *list1, *list2, *list3, *cases, *pos2, *counts;
  int32_t S, N;
  /* > 0: rvs_nm_run returns at its first look that finds at most this many simplices
   * running (none parked); a second call with the same state (and 0 here) runs the
   * rest -- every call starts by testing all simplices and listing the running ones,
   * so the rounds continue where they stopped, bit for bit.  What a caller gains: the
   * spectra that are done can go on while the stragglers' last rounds run. */
  int32_t stop_below, reserved_;
} rvs_nm_state;
/* One arm's MLP evaluator for rvs_nm_run: the arguments of rvs_template_nn and
 * rvs_nn_outside (xeqs == NULL: no hull, outside = 0), and the buffers the
 * round's template rows / outside flags go to ([>= S, ntp] / [>= S], device). */
typedef struct rvs_nm_nn_arm {
  const double *M, *S;
  const float *const *W, *const *b; /* HOST arrays of nlayer device pointers */
  const int32_t *dims;              /* HOST [nlayer + 1] */
  float *act0, *act1;
  double *templ, *outside;
  const double *xeqs, *yeqs;
  int32_t nlayer, nfx, nfy;
  uint32_t log_mask;
} rvs_nm_nn_arm;

/* rvs_template_nn + rvs_nn_outside of narm MLPs at the same B rows of `params`
 * (an optimiser round: a few hundred rows per arm, where three dependent launch
 * chains cost more than their arithmetic): three launches in all when the
 * networks have one shape up to the output width, else arm by arm.  Results in
 * arms[a].templ / .outside; same values as the per-arm calls. */
int rvs_template_nn_arms(const double *params, int B, int ndim, int narm,
                         const rvs_nm_nn_arm *arms, void *stream);
/* ... of the first min(B, njobs_dev[0]) rows only (a count on the device, as in
 * rvs_objective_fused_n; the grouped launches honour it, the arm-by-arm fallback
 * evaluates all B rows).  NULL = all B. */
int rvs_template_nn_arms_n(const double *params, int B, const int32_t *njobs_dev,
                           int ndim, int narm, const rvs_nm_nn_arm *arms,
                           void *stream);

/* One arm's Delaunay evaluator for rvs_nm_run / rvs_bfgs_run: the arguments of
 * rvs_template_tri_buckets and the buffers the round's template rows / outside flags /
 * simplex ids go to ([>= rows, ntp] / [>= rows] / int32 [>= rows], device). */
typedef struct rvs_nm_tri_arm {
  const double *dats, *transform, *extraflags;
  const int32_t *simplices;
  double *templ, *outside;
  int32_t *simplex;
  rvs_tri_buckets buckets;
  int32_t ntp, nsimplex, exp_flag;
  uint32_t log_mask;
} rvs_nm_tri_arm;

typedef struct rvs_nm_objective {
  const rvs_objective_arm *arms;
  const double *fixed, *vsini_fixed, *safe, *prior_mean, *prior_isig;
  double *vel, *vsini, *params, *extra, *chi;
  int32_t *job_spec, *bad, *jstatus, *status;
  void *scratch;
  double min_vel, max_vel, max_vsini, badchi;
  int32_t narm, npoly, n, ndim, vsini_col;
  int32_t src[8];
  /* NULL: regular-grid libraries, rvs_objective_fused.  Else [narm]: MLP
   * libraries -- a round's objective is rvs_template_nn + rvs_nn_outside per
   * arm, then rvs_objective_from_template (vel_fit.py:505-737 with
   * nn/RVSInterpolator.py:36-71 as the evaluator) */
  const rvs_nm_nn_arm *nn;
  /* NULL, or [narm]: Delaunay libraries on every arm (spec_inter.py:11-59) -- a
   * round's objective is rvs_template_tri_buckets per arm, then
   * rvs_objective_from_template.  At most one of nn / tri is set. */
  const rvs_nm_tri_arm *tri;
} rvs_nm_objective;
int rvs_nm_run(const rvs_nm_state *m, const rvs_nm_objective *o, double xatol,
               double fatol, int maxiter, int sync_every, int64_t *stats,
               void *stream);

/* The second minimiser of vel_fit.process (vel_fit.py:653-658: scipy's BFGS from the
 * simplex optimum, `second_minimizer`, the default of utils.py:26) with its rounds
 * on the device: the S runs are the state machines of rvs_bfgs_begin / _pending /
 * _feed (one source, csrc/bfgs_machine.h), one thread per spectrum; the rows they
 * ask for -- a value, the n forward-difference points of a gradient, or both -- are
 * gathered into one list per round and evaluated by the objective of rvs_nm_run in
 * chunks of `cap` rows (the row capacity of o's buffers), the row counts staying on
 * the device.  Blocks the calling host thread until every run has ended.
 *   runs      S * rvs_bfgs_run_bytes() bytes of device memory (8-byte aligned)
 *   x0 [S, n] start points, hess_inv0 [n, n] (device; NULL = identity)
 *   x [S, n], fun [S], hess_inv [S, n, n] (nullable), nit / nfev / status [S]
 *             (scipy's warnflag): results
 *   nreq, off [S]; list [S (n + 1)]; X [S (n + 1), n]; F [S (n + 1)]; counts [32]:
 *             work arrays (list zero-filled by the caller)
 *   n = o->n <= 8; S (n + 1) <= 24 cap; maxiter <= 0: 200 n
 * stats (nullable) int64[3] = rounds, objective calls, rows launched. */
typedef struct rvs_bfgs_state {
  void *runs;
  const double *x0, *hess_inv0;
  double *x, *fun, *hess_inv;
  int32_t *nit, *nfev, *status;
  int32_t *nreq, *off, *list, *counts;
  double *X, *F;
  double gtol, c1, c2, xrtol;
  int32_t S, n, cap, maxiter;
} rvs_bfgs_state;
int64_t rvs_bfgs_run_bytes(void);
int rvs_bfgs_run(const rvs_bfgs_state *b, const rvs_nm_objective *o,
                 int sync_every, int64_t *stats, void *stream);

#ifdef __cplusplus
}
#endif
#endif
