import ctypes, sys, os, json
import numpy as np
sys.path.insert(0, os.getcwd())
from mpp_cnn_rs_object_detection_amd import hip_api
hip_api.LIB_PATH = os.path.join(os.getcwd(), 'mpp_cnn_rs_object_detection_amd', 'libmppgpu_prof.so')
L = hip_api.load_library(hip_api.LIB_PATH); hip_api._lib = L
import bench
from mpp_cnn_rs_object_detection_amd import kernels, mappings, synth
setup, model = bench.load_model(); maps = mappings.default_mappings()
t = synth.make_tile(512, 200, 0)
for spec, lanes in ((1, 0), (8, 0)):
    ctx = hip_api.MppContext(0, point_capacity=1024, spec_waves=spec, spec_lanes=lanes)
    ctx.set_maps(t.det, t.marks); ctx.set_model(model, maps); ctx.naive_init(setup.detection_threshold, 6.0)
    xy, mk = ctx.get_points(); ctx.set_kernels(kernels.make_kernels(maps, float(len(xy))))
    ctx.set_schedule(1.0, 0.999, 0.0)
    ctx.run(100001, seed=0)
    buf = (ctypes.c_ulonglong * 16)(); L.mpp_debug_read_prof(buf)
    v = np.array(list(buf), dtype=float); names = ['draw','evaluate','commit','sync','dens','geom','unit','evalD','green','barrier_wait']
    buf2 = (ctypes.c_ulonglong * 16)(); L.mpp_debug_read_prof2(buf2, 1)
    print('   evalD parts (setup, after-pass, cand-loads, overlap-phase, combine, pair-loop, finish+stash):', [round(x/100001) for x in list(buf2)[:7]], 'clips/step', buf2[8]/100001, 'cands/eval', buf2[9]/max(1,buf2[11]), 'rescans/step', buf2[10]/100001, 'evals/step', buf2[11]/100001, 'zero-area clips/step', buf2[12]/100001)
    print('   data-driven birth draw (row search, column search, det + mark rows + marks), cycles per birth:', [round(list(buf2)[i] / max(1, 100001 / 9 / (8 if spec == 8 else 1))) for i in (13, 14, 15)])
    buf3 = (ctypes.c_ulonglong * 16)(); L.mpp_debug_read_prof3(buf3)
    buf4 = (ctypes.c_ulonglong * 16)(); L.mpp_debug_read_prof4(buf4)
    print('   draw cycles by kernel (UB,UD,DB,DD,GT,DT,GTF,DTF):', [round(buf4[k]/max(1,buf3[8+k])) for k in range(8)])
    print('   evaluate cycles by kernel (UB,UD,DB,DD,GT,DT,GTF,DTF):', [round(buf3[k]/max(1,buf3[8+k])) for k in range(8)], 'counts', list(buf3)[8:16])
    if spec == 8:
        sb = (ctypes.c_ulonglong * 64)(); L.mpp_debug_read_strag(sb); g = list(sb); R = max(1, g[24])
        print('   rounds', g[24], 'mean of round max', round(g[25] / R), 'mean of round mean', round(g[26] / R), 'rounds whose slowest wave re-reduced', round(g[27] / R, 3))
        print('   slowest wave by kernel (UB,UD,DB,DD,GT,DT,GTF,DTF): share', [round(g[k] / R, 3) for k in range(8)], 'its time', [round(g[8 + k] / max(1, g[k])) for k in range(8)], 'lead over 2nd', [round(g[16 + k] / max(1, g[k])) for k in range(8)])
        print('   step time by kernel, all waves:', [round(g[36 + k] / max(1, g[28 + k])) for k in range(8)], 'share of steps', [round(g[28 + k] / max(1, sum(g[28:36])), 3) for k in range(8)], 're-reducing steps: share', round(g[44] / max(1, sum(g[28:36])), 3), 'time', round(g[45] / max(1, g[44])))
    print('spec', spec, 'lanes', lanes, 'kernel ms', ctx.last_kernel_ms(), {n: round(x/100001) for n, x in zip(names, v)}, 'sum', round(v.sum()/100001), 'clock64 ticks/step')
