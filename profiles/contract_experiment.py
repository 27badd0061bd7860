#!/usr/bin/env python3
"""What FMA contraction would change: run 100 TGV steps on the device with the library named by
FASTS_HIP_LIB (e.g. one built with -ffp-contract=fast) and print, every 10 steps, the largest
relative deviation per variable from the contraction-free CPU restatement.

    FASTS_HIP_LIB=/path/to/lib.so python profiles/contract_experiment.py [n] [scheme]
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))

import numpy as np                                   # noqa: E402
from fast_amd import FastC, Internal                 # noqa: E402
import fast_amd.PyTree as FastS                      # noqa: E402
from oracle import oracle as O                       # noqa: E402
from common import make_case, zone_arrays, rel_err   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
scheme = sys.argv[2] if len(sys.argv) > 2 else 'senseur'
t = make_case('tgv', n, scheme, 'NSLaminar')
(t, tc, metrics) = FastS.warmup(t, None)
z = Internal.getZones(t)[0]
pint, preal, _ = zone_arrays(z)
oz = O.OracleZone(pint, preal, FastC.level_buffer(z).copy(), metrics[0][0])
out = []
for it in range(1, 101):
    FastS._compute(t, metrics, it, None)
    oz.step()
    if it % 10 == 0 or it == 1:
        e = rel_err(FastC.level_buffer(z), oz.ro, int(pint[41]))
        out.append({'step': it, 'max_rel_err': [float(x) for x in e]})
        print(it, e, flush=True)
print(json.dumps({'lib': os.environ.get('FASTS_HIP_LIB', 'default'), 'n': n, 'scheme': scheme, 'history': out}))
