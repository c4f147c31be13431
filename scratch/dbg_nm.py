import sys, numpy as np, torch
sys.path.insert(0, '.')
from rvspecfit_amd import neldermead, optimizer
rng = np.random.RandomState(3)
for S, N, maxiter, sync in ((700, 6, 10000, 4), (64, 5, 40, 1), (1500, 2, 10000, 7)):
    A = rng.normal(size=(S, N, N)); A = np.einsum('sij,skj->sik', A, A) + np.eye(N)
    At = torch.as_tensor(A).to('cuda'); ct = torch.as_tensor(rng.normal(size=(S, N))).to('cuda')
    def f(idx, X):
        d = X - ct[idx]; Ai = At[idx]; q = torch.zeros_like(d[:, 0])
        for i in range(N):
            for k in range(N):
                q = q + d[:, i] * Ai[:, i, k] * d[:, k]
            q = q + 3.0 * d[:, i].abs() + 2.0 * torch.sin(5 * d[:, i]).abs()
        return torch.where(X[:, 0] > 4.0, torch.full_like(q, 1e30), q)
    simp = torch.as_tensor(rng.normal(size=(S, N + 1, N)) * 2).to('cuda')
    r0 = neldermead.minimize(f, simp, maxiter=maxiter)
    r1 = optimizer.DeviceNelderMead(S, N, 'cuda').minimize(optimizer.TorchObjective(f), simp, maxiter=maxiter, sync_every=sync)
    print(S, N, 'nit eq', torch.equal(r0['nit'], r1['nit']), 'nfev eq', torch.equal(r0['nfev'], r1['nfev']),
          'succ', torch.equal(r0['success'], r1['success']))
    d = (r0['final_simplex'][0] - r1['final_simplex'][0]).abs().reshape(S, -1).max(1)[0]
    df = (r0['final_simplex'][1] - r1['final_simplex'][1]).abs().max(1)[0]
    bad = torch.nonzero((d > 0) | (df > 0)).reshape(-1)
    print(' rows differing', bad.numel(), 'max dx', float(d.max()), 'max df', float(df.max()))
    if bad.numel():
        i = int(bad[0]); print(i, r0['nit'][i], r1['nit'][i], r0['nfev'][i], r1['nfev'][i]);
        print(r0['final_simplex'][1][i]); print(r1['final_simplex'][1][i])
        print(r0['final_simplex'][0][i] - r1['final_simplex'][0][i])
