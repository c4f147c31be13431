import cProfile, pstats, sys, io
sys.argv = ['x', '300']
src = open('scratch/proc_time.py').read().replace('for it in range(2):', 'for it in range(1):')
pr = cProfile.Profile()
pr.enable()
exec(compile(src, 'proc_time', 'exec'))
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])
