import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from test_gpu_chain import setup_case
for spec in (1, 4):
    t, o, c = setup_case(128, 40, "legacy", spec=spec)
    c.set_schedule(1.0, 0.9985, 0.0)
    out, props = c.run(14, 7, trace_tile=0)
    print('spec', spec, 'n0', 'lds', c.get_option('lds_bytes'))
    for i in range(14):
        print(i, props[i]['kernel'], props[i]['target'], props[i]['ax'], props[i]['ay'], round(float(props[i]['as']),3), out[i]['accepted'], out[i]['n_after'], round(float(out[i]['dE']),5))
