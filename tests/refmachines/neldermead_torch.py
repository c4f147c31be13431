"""Lock-step batched Nelder-Mead: S independent simplices advance together so that
every objective evaluation is ONE batched launch set over the spectra that need it.

The state machine is the one `vel_fit.process` runs per spectrum through
`scipy.optimize.minimize(method='Nelder-Mead', options=dict(fatol, xatol,
initial_simplex, maxiter, maxfev=inf))` (vel_fit.py:627-637); scipy's algorithm
(`scipy/optimize/_optimize.py::_minimize_neldermead`, non-adaptive, no bounds:
rho=1, chi=2, psi=0.5, sigma=0.5, stable ordering of the vertices, termination
test BEFORE each iteration) is restated branch for branch, so a simplex that
sees the same function values takes the same path and ends on the same vertices
(tests/test_tools_cpu.py checks that against scipy itself).

Nothing here touches the GPU directly: the tensors live wherever the objective
puts them; the only host synchronisation per iteration is the compaction of the
still-active spectra.
"""
import torch

RHO, CHI, PSI, SIGMA = 1.0, 2.0, 0.5, 0.5


def _order(sim, fsim):
    # np.argsort on <= 16 elements is an insertion sort, i.e. stable; NaN last
    key = torch.where(torch.isnan(fsim), torch.full_like(fsim, float('inf')),
                      fsim)
    ind = torch.sort(key, dim=1, stable=True)[1]
    fsim = torch.gather(fsim, 1, ind)
    sim = torch.gather(sim, 1, ind[:, :, None].expand_as(sim))
    return sim, fsim


def minimize(func, simplex, fatol=1e-3, xatol=1e-2, maxiter=10000,
             stats=None):
    """simplex [S, N+1, N]; func(idx [J] long, X [J, N]) -> f [J] float64.

    Returns dict(x [S,N], fun [S], nit [S], nfev [S], success [S] bool,
    final_simplex (sim [S,N+1,N], fsim [S,N+1]))."""
    sim = simplex.clone().to(torch.float64)
    S, Np1, N = sim.shape
    assert Np1 == N + 1
    dev = sim.device
    allidx = torch.arange(S, device=dev)
    fsim = torch.empty((S, Np1), dtype=torch.float64, device=dev)
    for k in range(Np1):
        fsim[:, k] = func(allidx, sim[:, k].contiguous())
    sim, fsim = _order(sim, fsim)
    nit = torch.ones(S, dtype=torch.int64, device=dev)
    nfev = torch.full((S, ), Np1, dtype=torch.int64, device=dev)
    active = torch.ones(S, dtype=torch.bool, device=dev)
    success = torch.zeros(S, dtype=torch.bool, device=dev)
    rounds = 0
    while True:
        dx = (sim[:, 1:] - sim[:, :1]).abs().reshape(S, -1).max(dim=1)[0]
        df = (fsim[:, :1] - fsim[:, 1:]).abs().max(dim=1)[0]
        active &= nit < maxiter  # scipy: while-condition before the test
        conv = active & (dx <= xatol) & (df <= fatol)
        success |= conv
        active &= ~conv
        idx = torch.nonzero(active).reshape(-1)
        J = idx.numel()
        if J == 0:
            break
        rounds += 1
        s = sim[idx]
        f = fsim[idx]
        xbar = s[:, 0].clone()
        for j in range(1, N):
            xbar = xbar + s[:, j]
        # a true division: torch's GPU kernel turns `tensor / python_scalar`
        # into a multiplication by the reciprocal, numpy does not
        xbar = xbar / torch.full_like(xbar, N)
        worst = s[:, -1]
        xr = (1 + RHO) * xbar - RHO * worst
        fxr = func(idx, xr.contiguous())
        nf = torch.ones(J, dtype=torch.int64, device=dev)
        c_exp = fxr < f[:, 0]
        c_acc = (~c_exp) & (fxr < f[:, -2])
        c_oc = (~c_exp) & (~c_acc) & (fxr < f[:, -1])
        c_ic = (~c_exp) & (~c_acc) & (~c_oc)
        # second point: expansion / outside contraction / inside contraction
        x2 = torch.where(
            c_exp[:, None], (1 + RHO * CHI) * xbar - RHO * CHI * worst,
            torch.where(c_oc[:, None],
                        (1 + PSI * RHO) * xbar - PSI * RHO * worst,
                        (1 - PSI) * xbar + PSI * worst))
        need2 = ~c_acc
        f2 = torch.full_like(fxr, float('inf'))
        j2 = torch.nonzero(need2).reshape(-1)
        if j2.numel():
            f2[j2] = func(idx[j2], x2[j2].contiguous())
            nf[j2] += 1
        take2 = (c_exp & (f2 < fxr)) | (c_oc & (f2 <= fxr)) | \
            (c_ic & (f2 < f[:, -1]))
        taker = (c_exp & ~(f2 < fxr)) | c_acc
        shrink = ~(take2 | taker)
        newx = torch.where(take2[:, None], x2, xr)
        newf = torch.where(take2, f2, fxr)
        rep = take2 | taker
        s[:, -1] = torch.where(rep[:, None], newx, s[:, -1])
        f[:, -1] = torch.where(rep, newf, f[:, -1])
        js = torch.nonzero(shrink).reshape(-1)
        if js.numel():
            ss = s[js]
            for j in range(1, Np1):
                ss[:, j] = ss[:, 0] + SIGMA * (ss[:, j] - ss[:, 0])
                f[js, j] = func(idx[js], ss[:, j].contiguous())
            s[js] = ss
            nf[js] += N
        s, f = _order(s, f)
        sim[idx] = s
        fsim[idx] = f
        nit[idx] += 1
        nfev[idx] += nf
    if stats is not None:
        stats['rounds'] = stats.get('rounds', 0) + rounds
    return dict(x=sim[:, 0].clone(), fun=fsim.min(dim=1)[0], nit=nit, nfev=nfev,
                success=success, final_simplex=(sim, fsim))
