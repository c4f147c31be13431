"""How many of the objective slots the lock-step Nelder-Mead launches are evaluations
scipy's algorithm needs: sum of nm_nfev against objective_evals (launched slots)."""
import sys
sys.argv = [sys.argv[0]] + sys.argv[1:]
src = open('tools/perf/proc_time.py').read().replace('for it in range(2):', 'for it in range(1):')
exec(compile(src, 'p', 'exec'))
nf = r['nm_nfev'].double()
print('scipy nfev: mean %.1f sum %d; launched slots %d; ratio %.3f' %
      (nf.mean().item(), int(nf.sum().item()), r['objective_evals'],
       r['objective_evals'] / nf.sum().item()))
