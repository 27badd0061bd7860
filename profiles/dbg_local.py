import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..')); sys.path.insert(0, os.path.join(HERE, '..', 'tests')); sys.path.insert(0, os.path.join(HERE, '..', 'tests', 'golden'))
from fast_amd import FastC, Internal
import fast_amd.PyTree as FastS
import make_golden_local as ML
name = sys.argv[1]
c = ML.CASES[name]
g = np.load(os.path.join(HERE, '..', 'tests', 'golden', name + '.npz'))
t = ML.build(c)
(t, tc, metrics) = FastS.warmup(t, None)
ctx = FastC.HOOK['ctx']
nssiter = int(g['pint0'][58])
np.set_printoptions(linewidth=200, precision=6)
for it in range(1, int(g['nit']) + 1):
    FastS._compute(t, metrics, it, None)
    if it % c['kmod'] == 0:
        h = g['hist_it%d_0' % it]
        l2, loo = ctx.residual_history(0, nssiter, h.shape[2])
        print('it', it, 'nbox', g['nbox_it%d' % it][:, 0])
        print(np.abs(l2 / h[:, 0] - 1).max(axis=1), np.abs(loo / h[:, 1] - 1).max(axis=1))
        print(np.abs(l2 / h[:, 0] - 1)); print(np.abs(loo / h[:, 1] - 1))
        print(loo[:, 5]); print(h[:, 1, 5])
