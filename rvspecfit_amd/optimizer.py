"""Device-resident optimiser stage of vel_fit.process.

`DeviceNelderMead` drives the rvs_nm_* kernels (csrc/nm.hip): the S simplices
and all their bookkeeping live in HBM, a round is a fixed sequence of launches
whose job counts are read on the device, and the host only looks at the counts
every `sync_every` rounds (to shrink its launch bound, to run parked shrinks and
to notice that everything has converged).  `ProcessObjective` is chisq_func of
vel_fit.py:229-254 as a fixed launch sequence on preallocated buffers:
rvs_proc_map -> per arm rvs_template_polylinear, rvs_vsini_convolve,
rvs_spline_construct -> rvs_chisq_point (all arms) -> rvs_proc_finish.

tests/refmachines/neldermead_torch.py is the same state machine in torch: the CPU
suite pins it to scipy (identical nit, nfev, final simplex), the GPU suite pins
these kernels to it bit for bit.  Its ~70 small torch calls and three host
synchronisations per round cost more than the GPU work, hence the kernels.
"""
import ctypes

import torch

from . import _lib


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class ProcessObjective:
    """chisq_func for rows (list[j], X[j]) on preallocated buffers."""

    def __init__(self, batch, libs, names, pd0, fixParam, fitVsini, config,
                 options, priors, safe_params, resols=None):
        from . import engine
        L = _lib.lib()
        self.L = L
        self.batch, self.libs = batch, libs
        dev = batch.device
        S = batch.S
        self.S, self.dev = S, dev
        self.npoly = options.get('npoly') or 5
        self.rbf = options.get('rbf_continuum', True)
        self.resols = resols
        self.ndim = len(names)
        f64 = dict(dtype=torch.float64, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        # parameter vector layout (vel, [vsini], free stellar parameters)
        k = 1
        self.vsini_col = -1
        self.has_vsini = 'vsini' in pd0
        if fitVsini:
            self.vsini_col = k
            k += 1
        src = []
        for x in names:
            if x in fixParam:
                src.append(-1)
            else:
                src.append(k)
                k += 1
        self.n = k
        self.src = (ctypes.c_int32 * self.ndim)(*src)
        self.fixed = torch.stack([pd0[_] for _ in names], dim=1).contiguous()
        self.vsini_fixed = pd0['vsini'].contiguous() if (
            self.has_vsini and not fitVsini) else None
        self.safe = safe_params.contiguous()
        self.prior_mean = self.prior_isig = None
        if priors:
            pm = torch.zeros((S, self.ndim), **f64)
            ps = torch.zeros((S, self.ndim), **f64)
            for i, x in enumerate(names):
                if x in priors:
                    m, sg = priors[x]
                    pm[:, i] = torch.as_tensor(m, **f64)
                    ps[:, i] = 1.0 / torch.as_tensor(sg, **f64)
            self.prior_mean, self.prior_isig = pm, ps
        self.min_vel, self.max_vel = float(config['min_vel']), float(
            config['max_vel'])
        self.max_vsini = float(config['max_vsini'])
        cap = S
        self.cap = cap
        self.job_spec = torch.zeros(cap, **i32)
        self.vel = torch.zeros(cap, **f64)
        self.vsini = torch.zeros(cap, **f64) if self.has_vsini else None
        self.params = torch.zeros((cap, self.ndim), **f64)
        self.extra = torch.zeros(cap, **f64)
        self.bad = torch.zeros(cap, **i32)
        self.chi = torch.zeros(cap, **f64)
        self.jstatus = torch.zeros(cap, **i32)
        self.status = torch.zeros(S, **i32)
        self.arm_buf = []
        narm = len(batch.arms)
        self.arr = (_lib.PointArm * narm)()
        self.badchi = float(batch.badchi)
        # (which form the objective takes decides which row buffers exist: the one-kernel
        # objective keeps no template or spline record in HBM, the from-template form
        # only the evaluator's rows -- 0.6 GB per 1000 rows less to allocate for the
        # sub-batches of vel_fit._post_nm)
        self.fused = engine.can_fuse_objective(batch, libs, resols,
                                               npoly=self.npoly)
        self.from_templ = (not self.fused) and engine.can_fuse_objective(
            batch, libs, resols, npoly=self.npoly, from_template=True)
        chain = not (self.fused or self.from_templ)
        for ia, arm in enumerate(batch.arms):
            lib = libs[arm.name]
            b = dict(templ=None if self.fused else torch.empty((cap, lib.ntp), **f64),
                     templ2=torch.empty((cap, lib.ntp), **f64)
                     if (self.has_vsini and chain) else None,
                     coef=torch.empty((cap, lib.ntp, 4), **f64) if chain else None,
                     outside=torch.empty(cap, **f64),
                     sx=torch.empty(cap, **i32),
                     nn=None if lib.kind != 'nn' else dict(
                         a0=torch.empty((cap, lib.nn_width()),
                                        dtype=torch.float32, device=dev),
                         a1=torch.empty((cap, lib.nn_width()),
                                        dtype=torch.float32, device=dev)),
                     pen=torch.empty(cap, **f64),
                     work=arm.work(lib, 0.0), polysT=arm.basis(self.npoly,
                                                               self.rbf))
            self.arm_buf.append(b)
            a = self.arr[ia]
            a.lam, a.polysT = arm.lam.data_ptr(), b['polysT'].data_ptr()
            a.spec, a.espec = arm.spec.data_ptr(), arm.espec.data_ptr()
            a.work, a.knots = b['work'].data_ptr(), lib.knots.data_ptr()
            a.coef = b['coef'].data_ptr() if chain else None
            a.penalty = b['pen'].data_ptr()
            a.npix, a.S, a.ntp = arm.npix, arm.S, lib.ntp
            engine.set_point_grid(a, arm, self.npoly)
            a.log_step = int(lib.log_step)
            # A9: the spectra's own resolution matrices or the resol_params
            # override, as engine.chisq_point wires them
            rs = engine._arm_resol(arm, ia, resols)
            if rs is not None:
                a.taps, a.taps_stride, a.nd = rs['taps'].data_ptr(), \
                    rs['stride'], rs['nd']
                self._resol_keep = getattr(self, '_resol_keep', []) + [rs]
        nb = L.rvs_chisq_point_work_size(cap, narm)
        self.scratch = torch.empty((nb + 7) // 8, **f64)
        # (from_templ: evaluators that are no grid gather (MLP, Delaunay) -- the
        # template rows of a round from their own kernel, everything behind them in one)
        if self.fused or self.from_templ:
            self.oarr = (_lib.ObjectiveArm * narm)()
            self._keep = engine.fill_objective_arms(self.oarr, batch, libs,
                                                    self.npoly, self.rbf, 0.0)
            nb = L.rvs_objective_work_size(cap, narm)
            self.oscratch = torch.empty((nb + 7) // 8, **f64)
        # MLP libraries on every arm: the rounds can run inside rvs_nm_run too
        self.nn_native = self.from_templ and all(
            libs[arm.name].kind == 'nn' for arm in batch.arms)
        # ... and Delaunay libraries (find_simplex through the bucket grid)
        self.tri_native = self.from_templ and all(
            libs[arm.name].kind == 'triangulation' and
            libs[arm.name]._tri_bk is not None for arm in batch.arms)
        self.streams = [torch.cuda.Stream(device=dev) for _ in batch.arms]
        self.ev_in = torch.cuda.Event()
        self.ev_out = [torch.cuda.Event() for _ in batch.arms]
        self.calls = 0
        self.jobs = 0

    def native_desc(self):
        """rvs_nm_objective: this objective for the C round driver"""
        o = _lib.NmObjective()
        o.arms = ctypes.addressof(self.oarr)
        for k, t in (('fixed', self.fixed), ('vsini_fixed', self.vsini_fixed),
                     ('safe', self.safe), ('prior_mean', self.prior_mean),
                     ('prior_isig', self.prior_isig), ('vel', self.vel),
                     ('vsini', self.vsini), ('params', self.params),
                     ('extra', self.extra), ('chi', self.chi),
                     ('job_spec', self.job_spec), ('bad', self.bad),
                     ('jstatus', self.jstatus), ('status', self.status),
                     ('scratch', self.oscratch)):
            setattr(o, k, None if t is None else t.data_ptr())
        o.min_vel, o.max_vel = self.min_vel, self.max_vel
        o.max_vsini, o.badchi = self.max_vsini, self.badchi
        o.narm, o.npoly = len(self.arm_buf), self.npoly
        o.n, o.ndim, o.vsini_col = self.n, self.ndim, self.vsini_col
        for i in range(8):
            o.src[i] = self.src[i] if i < self.ndim else -1
        o.nn = None
        if self.nn_native:
            narm = len(self.arm_buf)
            self._nn_arr = (_lib.NmNNArm * narm)()
            self._nn_keep = []
            for ia, (arm, b) in enumerate(zip(self.batch.arms, self.arm_buf)):
                lib = self.libs[arm.name]
                a = self._nn_arr[ia]
                nl = len(lib.nn_W)
                Wp = (ctypes.c_void_p * nl)(*[w.data_ptr() for w in lib.nn_W])
                bp = (ctypes.c_void_p * nl)(*[x.data_ptr() for x in lib.nn_b])
                self._nn_keep += [Wp, bp]
                a.M, a.S = lib.nn_M.data_ptr(), lib.nn_S.data_ptr()
                a.W = ctypes.cast(Wp, ctypes.c_void_p)
                a.b = ctypes.cast(bp, ctypes.c_void_p)
                a.dims = lib.nn_dims.ctypes.data
                a.act0 = b['nn']['a0'].data_ptr()
                a.act1 = b['nn']['a1'].data_ptr()
                a.templ, a.outside = b['templ'].data_ptr(), b['outside'].data_ptr()
                hull = lib.hull_device()
                if hull is None:
                    a.xeqs = a.yeqs = None
                    a.nfx = a.nfy = 0
                else:
                    a.xeqs, a.yeqs = hull[0].data_ptr(), hull[1].data_ptr()
                    a.nfx, a.nfy = hull[0].shape[0], hull[1].shape[0]
                a.nlayer, a.log_mask = nl, lib.log_mask
            o.nn = ctypes.addressof(self._nn_arr)
        o.tri = None
        if self.tri_native:
            narm = len(self.arm_buf)
            self._tri_arr = (_lib.NmTriArm * narm)()
            for ia, (arm, b) in enumerate(zip(self.batch.arms, self.arm_buf)):
                lib = self.libs[arm.name]
                a = self._tri_arr[ia]
                a.dats, a.transform = lib.dats.data_ptr(), lib.tri_transform.data_ptr()
                a.extraflags = lib.tri_extraflags.data_ptr()
                a.simplices = lib.tri_simplices.data_ptr()
                a.templ, a.outside = b['templ'].data_ptr(), b['outside'].data_ptr()
                # (arms on ONE triangulation share the simplex ids: rvs_nm_run searches
                # once for all of them)
                first = [k for k in range(ia + 1) if self.libs[
                    self.batch.arms[k].name].tri_transform.data_ptr() ==
                    lib.tri_transform.data_ptr() and self.libs[
                    self.batch.arms[k].name].log_mask == lib.log_mask][0]
                a.simplex = self.arm_buf[first]['sx'].data_ptr()
                a.buckets = lib._tri_bk
                a.ntp, a.nsimplex = lib.ntp, lib.tri_nsimplex
                a.exp_flag, a.log_mask = lib.exp_flag, lib.log_mask
            o.tri = ctypes.addressof(self._tri_arr)
        return o

    def eval(self, list_t, X, J, counts, cidx, F):
        """F[:J] = chisq_func(X[j]) for spectrum list_t[j]; rows >= the device
        count counts[cidx] are padding: the fused objective skips them, F there
        is whatever it was."""
        L = self.L
        st = _lib.stream()
        rc = L.rvs_proc_map(J, self.n, self.ndim, _p(X), _p(list_t), self.src,
                            self.vsini_col, _p(self.fixed),
                            _p(self.vsini_fixed), _p(self.safe),
                            _p(self.prior_mean), _p(self.prior_isig),
                            self.min_vel, self.max_vel, self.max_vsini,
                            _p(self.job_spec), _p(self.vel), _p(self.vsini),
                            _p(self.params), _p(self.extra), _p(self.bad), st)
        _lib.check(rc, 'rvs_proc_map')
        if self.fused:   # one kernel: gather, FIR, spline solve, chi^2
            # 1 | RVS_OBJ_STATUS_STORE: jstatus is overwritten, no clearing launch
            live = None if counts is None else \
                counts.data_ptr() + 4 * int(cidx)
            rc = L.rvs_objective_fused_n(
                ctypes.addressof(self.oarr), len(self.arm_buf), self.npoly,
                _p(self.params), _p(self.vsini), _p(self.job_spec), J, live,
                _p(self.vel), self.badchi, 3, _p(self.oscratch), _p(self.chi),
                _p(self.jstatus), st)
            _lib.check(rc, 'rvs_objective_fused')
            rc = L.rvs_proc_finish(J, _p(counts), cidx, _p(self.chi),
                                   _p(self.extra), _p(self.bad),
                                   _p(self.job_spec), _p(self.jstatus), _p(F),
                                   _p(self.status), st)
            _lib.check(rc, 'rvs_proc_finish')
            self.calls += 1
            self.jobs += J
            return
        if self.from_templ and self.nn_native:
            # MLP libraries on every arm: one grouped launch chain
            if getattr(self, '_nn_arr', None) is None:
                self.native_desc()
            rc = L.rvs_template_nn_arms_n(
                _p(self.params), J,
                None if counts is None else counts.data_ptr() + 4 * int(cidx),
                self.ndim, len(self.arm_buf), ctypes.addressof(self._nn_arr), st)
            _lib.check(rc, 'rvs_template_nn_arms')
        elif self.from_templ:
            main = torch.cuda.current_stream()
            self.ev_in.record(main)
            for arm, b, side, ev in zip(self.batch.arms, self.arm_buf,
                                        self.streams, self.ev_out):
                lib = self.libs[arm.name]
                side.wait_event(self.ev_in)
                scr = b['sx']
                if b['nn'] is not None:
                    scr = dict(b['nn'], torch_stream=side)
                lib.eval_into(self.params, J, b['templ'], b['outside'],
                              ctypes.c_void_p(side.cuda_stream), scratch=scr)
                ev.record(side)
            for ev in self.ev_out:
                main.wait_event(ev)
        if self.from_templ:
            narm = len(self.arm_buf)
            tp = (ctypes.c_void_p * narm)(*[b['templ'].data_ptr()
                                            for b in self.arm_buf])
            op = (ctypes.c_void_p * narm)(*[b['outside'].data_ptr()
                                            for b in self.arm_buf])
            live = None if counts is None else \
                counts.data_ptr() + 4 * int(cidx)
            rc = L.rvs_objective_from_template_n(
                ctypes.addressof(self.oarr), narm, self.npoly,
                ctypes.cast(tp, ctypes.c_void_p),
                ctypes.cast(op, ctypes.c_void_p), _p(self.vsini),
                _p(self.job_spec), J, live, _p(self.vel), self.badchi, 3,
                _p(self.oscratch), _p(self.chi), _p(self.jstatus), st)
            _lib.check(rc, 'rvs_objective_from_template')
            rc = L.rvs_proc_finish(J, _p(counts), cidx, _p(self.chi),
                                   _p(self.extra), _p(self.bad),
                                   _p(self.job_spec), _p(self.jstatus), _p(F),
                                   _p(self.status), st)
            _lib.check(rc, 'rvs_proc_finish')
            self.calls += 1
            self.jobs += J
            return
        # the arms are independent until the point kernel: one stream each
        main = torch.cuda.current_stream()
        self.ev_in.record(main)
        for arm, b, side, ev in zip(self.batch.arms, self.arm_buf, self.streams,
                                    self.ev_out):
            lib = self.libs[arm.name]
            side.wait_event(self.ev_in)
            ss = ctypes.c_void_p(side.cuda_stream)
            scr = b['sx']
            if b['nn'] is not None:
                scr = dict(b['nn'], torch_stream=side)
            lib.eval_into(self.params, J, b['templ'], b['outside'], ss,
                          scratch=scr)
            y = b['templ']
            if self.has_vsini:
                rc = L.rvs_vsini_convolve(_p(y), _p(self.vsini), _p(b['outside']),
                                          lib.lnstep, 0.6, lib.ntp, J,
                                          _p(b['templ2']), ss)
                _lib.check(rc, 'rvs_vsini_convolve')
                y = b['templ2']
            rc = L.rvs_spline_construct(_p(lib.knots), _p(y), lib.ntp, J,
                                        lib.spline_form, _p(lib.spline_factors),
                                        _p(b['coef']), ss)
            _lib.check(rc, 'rvs_spline_construct')
            with torch.cuda.stream(side):
                torch.mul(b['outside'], self.badchi, out=b['pen'])
                if self.batch.pen_scale is not None:   # grid sets (engine.SpecBatch)
                    b['pen'][:J] *= self.batch.pen_scale[self.job_spec[:J].long()]
            ev.record(side)
        for ev in self.ev_out:
            main.wait_event(ev)
        from . import engine
        if self.npoly > engine.POINT_MAXP:
            # 17 ... 32 basis functions: beyond the point kernel's 16 per lane, the
            # arms' values come from rvs_chisq_full (engine.chisq_point)
            c, stj = engine.chisq_point(
                self.batch, self.libs, [b['coef'] for b in self.arm_buf],
                [b['outside'][:J] for b in self.arm_buf], self.vel[:J],
                npoly=self.npoly, rbf=self.rbf, job_spec=self.job_spec[:J],
                resols=self.resols)
            self.chi[:J] = c
            self.jstatus[:J] = stj
        else:
            self.jstatus.zero_()
            rc = L.rvs_chisq_point(ctypes.addressof(self.arr), len(self.arm_buf),
                                   self.npoly, _p(self.job_spec), None, J,
                                   _p(self.vel), self.badchi, _p(self.scratch),
                                   _p(self.chi), _p(self.jstatus), st)
            _lib.check(rc, 'rvs_chisq_point')
        rc = L.rvs_proc_finish(J, _p(counts), cidx, _p(self.chi), _p(self.extra),
                               _p(self.bad), _p(self.job_spec), _p(self.jstatus),
                               _p(F), _p(self.status), st)
        _lib.check(rc, 'rvs_proc_finish')
        self.calls += 1
        self.jobs += J


# False: the rounds of a fused objective are driven from Python (one_round below,
# the loop every non-fused objective takes anyway) instead of rvs_nm_run; the
# two give the same simplices bit for bit
# (tests/test_gpu_parity.py::test_nm_round_drivers_agree).
# (Replaying the rounds from HIP graphs was measured in round 1: same wall time --
# the rounds are bound by the GPU-side chain of small dependent kernels, not by
# host launches -- and removed.)
NATIVE_ROUNDS = True


def _order(sim, fsim):
    """scipy: ind = np.argsort(fsim); sim = np.take(sim, ind, 0) -- stable, so
    that equal values keep their vertex order"""
    # (np.argsort on <= 16 elements is an insertion sort, i.e. stable; NaN last)
    key = torch.where(torch.isnan(fsim), torch.full_like(fsim, float('inf')),
                      fsim)
    ind = torch.sort(key, dim=1, stable=True)[1]
    fsim = torch.gather(fsim, 1, ind)
    sim = torch.gather(sim, 1, ind[:, :, None].expand_as(sim))
    return sim, fsim


class DeviceNelderMead:

    def __init__(self, S, N, dev):
        self.S, self.N, self.dev = S, N, dev
        f64 = dict(dtype=torch.float64, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.fsim = torch.empty((S, N + 1), **f64)
        self.nit = torch.ones(S, **i32)
        self.nfev = torch.full((S, ), N + 1, **i32)
        self.flags = torch.ones(S, **i32)
        self.list1 = torch.zeros(S, **i32)
        self.list2 = torch.zeros(S, **i32)
        self.list3 = torch.zeros(S, **i32)
        self.X1 = torch.zeros((S, N), **f64)
        self.X2 = torch.zeros((S, N), **f64)
        self.F1 = torch.zeros(S, **f64)
        self.F2 = torch.zeros(S, **f64)
        self.cases = torch.zeros(S, **i32)
        self.pos2 = torch.zeros(S, **i32)
        self.counts = torch.zeros(8, **i32)

    def minimize(self, objective, simplex, fatol=1e-3, xatol=1e-2,
                 maxiter=10000, sync_every=4, stats=None, stop_below=0):
        """stop_below > 0 (rounds inside the library only): return at the first look
        that finds at most that many simplices running -- the result then carries
        paused = True and finished [S] (converged and out of the rounds); resume()
        runs the rest."""
        L = _lib.lib()
        S, N = self.S, self.N
        sim = simplex.clone().to(torch.float64).contiguous()
        allidx = torch.arange(S, dtype=torch.int32, device=self.dev)
        for k in range(N + 1):
            self.X1.copy_(sim[:, k])
            objective.eval(allidx, self.X1, S, None, 0, self.F1)
            self.fsim[:, k] = self.F1
        sim, fsim = _order(sim, self.fsim)
        sim = sim.contiguous()
        self.fsim.copy_(fsim)
        fs = self.fsim
        if NATIVE_ROUNDS and isinstance(objective, ProcessObjective) and \
                (objective.fused or objective.nn_native or objective.tri_native):
            # the rounds in C (rvs_nm_run): same launches, no interpreter
            m = _lib.NmState()
            for k, t in (('sim', sim), ('fsim', fs), ('X1', self.X1),
                         ('X2', self.X2), ('F1', self.F1), ('F2', self.F2),
                         ('nit', self.nit), ('nfev', self.nfev),
                         ('flags', self.flags), ('list1', self.list1),
                         ('list2', self.list2), ('list3', self.list3),
                         ('cases', self.cases), ('pos2', self.pos2),
                         ('counts', self.counts)):
                setattr(m, k, t.data_ptr())
            m.S, m.N = S, N
            m.stop_below = int(stop_below)
            o = objective.native_desc()
            self._native = (m, o, sim, fs, float(xatol), float(fatol), int(maxiter),
                            int(sync_every))
            return self._run_native(objective, stats)

        def one_round(jb):
            st = _lib.stream()
            rc = L.rvs_nm_begin(S, N, xatol, fatol, maxiter, _p(sim), _p(fs),
                                _p(self.nit), _p(self.flags), _p(self.list1),
                                _p(self.X1), _p(self.counts), jb, st)
            _lib.check(rc, 'rvs_nm_begin')
            objective.eval(self.list1, self.X1, jb, self.counts, 0, self.F1)
            rc = L.rvs_nm_decide(N, _p(sim), _p(fs), _p(self.list1),
                                 _p(self.F1), _p(self.cases), _p(self.pos2),
                                 _p(self.list2), _p(self.X2), _p(self.counts), jb,
                                 st)
            _lib.check(rc, 'rvs_nm_decide')
            objective.eval(self.list2, self.X2, jb, self.counts, 1, self.F2)
            rc = L.rvs_nm_update(N, _p(sim), _p(fs), _p(self.nit),
                                 _p(self.nfev), _p(self.list1), _p(self.X1),
                                 _p(self.F1), _p(self.cases), _p(self.pos2),
                                 _p(self.X2), _p(self.F2), _p(self.flags),
                                 _p(self.counts), jb, st)
            _lib.check(rc, 'rvs_nm_update')

        # launch bounds are quantised (1/8 steps of a power of two) so that a
        # handful of launch shapes serve the whole run
        def bucket(n):
            if n <= 64:
                return min(S, 64)
            p2 = 1 << (int(n - 1).bit_length())      # next power of two >= n
            stepq = max(p2 // 8, 1)
            return min(S, -(-n // stepq) * stepq)

        jb = S
        rounds = 0
        rc = L.rvs_nm_begin(S, N, xatol, fatol, maxiter, _p(sim), _p(fs),
                            _p(self.nit), _p(self.flags), _p(self.list1),
                            _p(self.X1), _p(self.counts), jb, _lib.stream())
        _lib.check(rc, 'rvs_nm_begin')
        while True:
            # host look: counts of the most recent begin (an upper bound of what
            # is active now), parked shrinks
            c = self.counts.cpu().numpy()
            live, parked = int(c[0]), int(c[4])
            if parked > 0:
                self._shrink(objective, sim, parked)
                rc = L.rvs_nm_begin(S, N, xatol, fatol, maxiter, _p(sim), _p(fs),
                                    _p(self.nit), _p(self.flags), _p(self.list1),
                                    _p(self.X1), _p(self.counts), S,
                                    _lib.stream())
                _lib.check(rc, 'rvs_nm_begin')
                continue
            if live == 0:
                break
            jb = bucket(live)
            for _ in range(sync_every):
                one_round(jb)
            rounds += sync_every
        if stats is not None:
            stats['rounds'] = stats.get('rounds', 0) + rounds
        success = (self.flags & 2) != 0
        return dict(x=sim[:, 0].clone(), fun=fs.min(dim=1)[0],
                    nit=self.nit.long(), nfev=self.nfev.long(), success=success,
                    final_simplex=(sim, fs))

    def _run_native(self, objective, stats):
        L = _lib.lib()
        m, o, sim, fs, xatol, fatol, maxiter, sync_every = self._native
        st3 = (ctypes.c_int64 * 3)()
        nfev0 = self.nfev.sum()
        rc = L.rvs_nm_run(ctypes.addressof(m), ctypes.addressof(o), xatol, fatol,
                          maxiter, sync_every, st3, _lib.stream())
        _lib.check(rc, 'rvs_nm_run')
        objective.calls += int(st3[1])
        # evaluations performed = the function values scipy's algorithm counts
        # (rows of a launch behind the device count are skipped); `slots` =
        # rows launched
        objective.jobs += int((self.nfev.sum() - nfev0).item())
        objective.slots = getattr(objective, 'slots', 0) + int(st3[2])
        if stats is not None:
            stats['rounds'] = stats.get('rounds', 0) + int(st3[0])
        success = (self.flags & 2) != 0
        out = dict(x=sim[:, 0].clone(), fun=fs.min(dim=1)[0],
                   nit=self.nit.long(), nfev=self.nfev.long(),
                   success=success, final_simplex=(sim, fs))
        if m.stop_below > 0:
            running = (self.flags & 5) != 0      # active, or parked for a shrink
            out['paused'] = bool(running.any().item())
            out['finished'] = (~running) & success
        return out

    def resume(self, objective, stats=None):
        """the rest of a run that minimize(stop_below > 0) returned from"""
        self._native[0].stop_below = 0
        return self._run_native(objective, stats)

    def _shrink(self, objective, sim, parked):
        """scipy's shrink step for the parked simplices: N objective calls"""
        L = _lib.lib()
        N = self.N
        st = _lib.stream()
        rc = L.rvs_nm_collect(self.S, _p(self.flags), _p(self.list3),
                              _p(self.counts), st)
        _lib.check(rc, 'rvs_nm_collect')
        jb = parked
        for k in range(1, N + 1):
            rc = L.rvs_nm_shrink_point(N, k, _p(sim), _p(self.list3),
                                       _p(self.X2), _p(self.counts), jb, st)
            _lib.check(rc, 'rvs_nm_shrink_point')
            objective.eval(self.list3, self.X2, jb, self.counts, 2, self.F2)
            rc = L.rvs_nm_shrink_store(N, k, _p(sim), _p(self.fsim),
                                       _p(self.nit), _p(self.nfev),
                                       _p(self.flags), _p(self.list3),
                                       _p(self.F2), _p(self.counts), jb, st)
            _lib.check(rc, 'rvs_nm_shrink_store')


class TorchObjective:
    """adapter: a torch callable f(idx long [J], X [J,N]) -> [J] behind the
    ProcessObjective.eval interface (tests of DeviceNelderMead)"""

    def __init__(self, func):
        self.func = func

    def eval(self, list_t, X, J, counts, cidx, F):
        F[:J] = self.func(list_t[:J].long(), X[:J])
