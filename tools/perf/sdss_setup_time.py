"""host + device time to build the per-grid tables of a 10 000-grid SDSS-shaped
batch (what bench.py --workload sdss does before its first step):
tools/perf/sdss_setup_time.py [S] [device_tables 0/1]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from rvspecfit_amd import engine
S = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
engine.DEVICE_TABLES = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
lam = bench.obs_lam('s')
pieces = bench.sdss_pieces(S, 3)
grids = [lam[a:a + n] for a, n in pieces]
npx = int(pieces[:, 1].max())
dev = torch.device('cuda', 0)
sp = torch.ones((S, npx), dtype=torch.float64, device=dev)
es = torch.full((S, npx), 0.1, dtype=torch.float64, device=dev)


class Lib:
    name = 'fake'
    cc = dict(npoints=8192, logl0=np.log(3750 / 1.0033), logl1=np.log(9300 * 1.0033),
              continuum=True, splinestep=13000.)

    def ccf_set(self, config):
        return self.cc


torch.cuda.synchronize(); t0 = time.time()
arm = engine.ArmData('s', grids, sp, es, device=dev, grid_id=np.arange(S, dtype=np.int32))
torch.cuda.synchronize(); t1 = time.time()
arm.basis(10, True); arm.basis_ortho(10, True)
torch.cuda.synchronize(); t2 = time.time()
arm.ccf_tables(Lib(), dict(max_vel=1000, vel_step0=5))
torch.cuda.synchronize(); t3 = time.time()
print('S %d device_tables %d: ArmData %.2f s, basis + ortho %.2f s, CCF tables %.2f s' % (
    S, engine.DEVICE_TABLES, t1 - t0, t2 - t1, t3 - t2))
