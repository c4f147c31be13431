"""cProfile of vel_fit.process at S spectra (second_minimizer on): the host side of a
call -- what Python does between the device stages.  usage: proc_host_profile.py [S]"""
import cProfile
import pstats
import sys
src = open('tools/perf/proc_time.py').read().split('for it in range(2):')[0]
exec(compile(src, 'proc_time_head', 'exec'))
cfg['second_minimizer'] = True
os.environ['RVS_PROCESS_STREAMS'] = os.environ.get('RVS_PROCESS_STREAMS', '1')
vel_fit.process(batch, pd0, options=bench.OPTIONS, config=cfg)      # warm
torch.cuda.synchronize()
tm = {}
t0 = time.time()
vel_fit.process(batch, pd0, options=bench.OPTIONS, config=cfg, timers=tm)
torch.cuda.synchronize()
print('S', S, 'timed stages (synchronised)', {k: round(v, 3) for k, v in tm.items()},
      'total %.3f' % (time.time() - t0))
t0 = time.time()
vel_fit.process(batch, pd0, options=bench.OPTIONS, config=cfg)
torch.cuda.synchronize()
print('untimed call %.3f s' % (time.time() - t0))
pr = cProfile.Profile()
pr.enable()
vel_fit.process(batch, pd0, options=bench.OPTIONS, config=cfg)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
st.sort_stats('cumulative').print_stats(40)
