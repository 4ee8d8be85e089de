"""debug: kernel dE (incremental) vs oracle vs the from-scratch GPU kernel wherever they differ by > 1e-9 (config 3 tiles)"""
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
seed = 20261004
for tile_id in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    t = synth.make_tile(512, 200, tile_id=tile_id)
    o = oracle.Oracle(t.det.shape, t.det, t.marks, model, kernels.make_kernels(maps, 1.0))
    xy0, mk0 = o.naive_detection(setup.detection_threshold, 6.0)
    kd = kernels.make_kernels(maps, float(max(1, len(xy0))))
    o = oracle.Oracle(t.det.shape, t.det, t.marks, model, kd)
    o.set_points(xy0, mk0); o.set_temperature(1.0, 0.999, 0.0)
    ctx = hip_api.MppContext(0, point_capacity=1024, spec_waves=8)
    ctx.set_maps(t.det, t.marks); ctx.set_model(model, maps)
    ctx.set_points(0, xy0, mk0); ctx.set_kernels(kd); ctx.set_schedule(1.0, 0.999, 0.0)
    done, total, nbad, nflip = 0, 100001, 0, 0
    while done < total:
        n = min(20000, total - done)
        gout, gprops = ctx.run(n, seed=seed, chain0=tile_id, trace_tile=0)
        saved = o.save()
        out = np.zeros(n, oracle.STEPOUT_DTYPE)
        import ctypes
        acc = np.ascontiguousarray(gout["accepted"], dtype=np.int32)
        tape = np.ascontiguousarray(gprops, dtype=oracle.PROPOSAL_DTYPE)
        # pass 1: forced replay recording the oracle's dE (its decision would be log(u) < log_alpha)
        oracle.lib().orc_replay_forced(o._h, n, tape.ctypes.data_as(ctypes.c_void_p), acc.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
        d = np.abs(out["dE"] - gout["dE"])
        bad = np.nonzero(d > 1e-9 * np.maximum(1, np.abs(out["dE"])))[0]
        own = (np.log(gprops["u_accept"] + 1e-16) < out["log_alpha"]).astype(np.int32)
        nflip += int((own != gout["accepted"]).sum())
        nbad += len(bad)
        for s in bad[:3]:
            o2 = oracle.Oracle(t.det.shape, t.det, t.marks, model, kd)
            o2.restore(saved, 0.999)
            if s:
                o2.replay_forced(gprops[:s], gout["accepted"][:s])
            sxy, sm = o2.get_points()
            p = gprops[s]
            rem = [int(p["target"])] if p["target"] >= 0 else []
            has_add = p["kernel"] not in (1, 3)
            axy = [[int(p["ax"]), int(p["ay"])]] if has_add else []
            am = [[p["as"], p["ar"], p["aa"]]] if has_add else []
            print(f"tile {tile_id} step {done + s}: kernel {p['kernel']} dE gpu {gout['dE'][s]!r} oracle(step) {out['dE'][s]!r} diff {d[s]:.3e} T {gout['T'][s]:.2e}")
            print("   oracle delta()", repr(o2.delta(rem, axy if axy else None, am if am else None)))
            c2 = hip_api.MppContext(0, point_capacity=1024)
            c2.set_maps(t.det, t.marks); c2.set_model(model, maps); c2.set_points(0, sxy, sm)
            print("   from-scratch GPU kernel (mpp_delta_batch)", repr(c2.delta_batch(0, [rem], [axy], [am])[0]))
            print("   proposal", p, "target rect", (sxy[p['target']].tolist(), [repr(v) for v in sm[p['target']].tolist()]) if rem else None)
            cx, cy = (p["ax"], p["ay"]) if has_add else sxy[p["target"]]
            dd = np.sqrt(((sxy - np.array([cx, cy])) ** 2).sum(1))
            for j in np.nonzero(dd <= 40)[0]:
                line = f"   neighbour slot {j} at {sxy[j].tolist()} marks {[repr(v) for v in sm[j].tolist()]} dist {dd[j]:.2f}"
                if has_add:
                    line += f" overlap(new,nb)={oracle.overlap([p['ax'], p['ay'], p['as'], p['ar'], p['aa']], [sxy[j][0], sxy[j][1], *sm[j]])!r}"
                if rem and j != p["target"]:
                    line += f" overlap(old,nb)={oracle.overlap([*sxy[p['target']], *sm[p['target']]], [sxy[j][0], sxy[j][1], *sm[j]])!r}"
                print(line)
            e_g, vec_g = c2.total_energy(0, return_vectors=True)
            e_o, vec_o = o2.total_energy(return_vectors=True)
            print("   total energy gpu-from-scratch", repr(e_g), "oracle", repr(e_o), "max |vec diff|", np.abs(vec_g - vec_o).max(), "argmax", np.unravel_index(np.abs(vec_g - vec_o).argmax(), vec_g.shape))
            c2.close()
        done += n
    print(f"tile {tile_id}: {nbad} steps with |dE diff| > 1e-9, {nflip} decisions the oracle would take differently", flush=True)
    ctx.close()
