"""debug: where do kernel and oracle dE differ by more than 1e-9 on tile 0 of config 3, and why"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import oracle
from helpers import hrc_model
from mpp_cnn_rs_object_detection_amd import energies as E, hip_api, kernels, mappings, synth

setup, comb = hrc_model()
unit, pair = setup.make_energies()
model = E.build_model_desc(unit, pair, comb)
maps = mappings.default_mappings()
t = synth.make_tile(512, 200, tile_id=0)
seed = 20261004
o = oracle.Oracle(t.det.shape, t.det, t.marks, model, kernels.make_kernels(maps, 1.0))
xy0, mk0 = o.naive_detection(setup.detection_threshold, 6.0)
kd = kernels.make_kernels(maps, float(max(1, len(xy0))))
o = oracle.Oracle(t.det.shape, t.det, t.marks, model, kd)
o.set_points(xy0, mk0); o.set_temperature(1.0, 0.999, 0.0)
ctx = hip_api.MppContext(0, point_capacity=1024, spec_waves=8)
ctx.set_maps(t.det, t.marks); ctx.set_model(model, maps)
ctx.set_points(0, xy0, mk0); ctx.set_kernels(kd); ctx.set_schedule(1.0, 0.999, 0.0)
done, total = 0, 100001
while done < total:
    n = min(1000, total - done)
    gxy, gm = ctx.get_points(0)            # state before the chunk
    gout, gprops = ctx.run(n, seed=seed, chain0=0, trace_tile=0)
    o.set_points(gxy, gm); 
    import ctypes
    oracle.lib().orc_set_step_index(o._h, done)
    oout = o.replay(gprops) if False else None
    # follow the kernel's tape with the kernel's decisions, recording the oracle's own dE
    out = np.zeros(n, oracle.STEPOUT_DTYPE)
    acc = np.ascontiguousarray(gout["accepted"], dtype=np.int32)
    tape = np.ascontiguousarray(gprops, dtype=oracle.PROPOSAL_DTYPE)
    oracle.lib().orc_replay_forced(o._h, n, tape.ctypes.data_as(ctypes.c_void_p), acc.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    d = np.abs(out["dE"] - gout["dE"])
    bad = np.nonzero(d > 1e-9 * np.maximum(1, np.abs(out["dE"])))[0]
    for s in bad:
        print(f"step {done + s}: kernel {gprops['kernel'][s]} target {gprops['target'][s]} dE gpu {gout['dE'][s]!r} oracle {out['dE'][s]!r} diff {d[s]:.3e} T {gout['T'][s]:.3e}")
        # rebuild the state before step s: replay the chunk's first s steps on a scratch oracle
        o2 = oracle.Oracle(t.det.shape, t.det, t.marks, model, kd)
        o2.set_points(gxy, gm); oracle.lib().orc_set_step_index(o2._h, done)
        if s:
            o2.replay_forced(gprops[:s], gout["accepted"][:s])
        sxy, sm = o2.get_points()
        p = gprops[s]
        tx, ty = (sxy[p["target"]] if p["target"] >= 0 else (p["ax"], p["ay"]))
        print("   proposal", p, " target rect", sxy[p["target"]].tolist() if p["target"] >= 0 else None, sm[p["target"]].tolist() if p["target"] >= 0 else None)
        dd = np.sqrt(((sxy - np.array([p["ax"], p["ay"]])) ** 2).sum(1))
        for j in np.nonzero(dd <= 34)[0]:
            ov_new = oracle.overlap([p["ax"], p["ay"], p["as"], p["ar"], p["aa"]], [sxy[j][0], sxy[j][1], *sm[j]])
            print(f"   neighbour slot {j} at {sxy[j].tolist()} marks {sm[j].tolist()} dist {dd[j]:.2f} overlap(new,nb)={ov_new!r}")
    done += n
print("done")
