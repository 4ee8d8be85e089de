#!/usr/bin/env python3
"""Bench of the MPP / RJMCMC sampling hot path on MI355X.

Workload (BASELINE.json configs[1]): ONE 512x512 synthetic tile with ~200 objects, mpp_hrcM
energies, the reference schedule T0=1, alpha=0.999, burn_in=99744 + 2*128 samples -> 100 001
RJMCMC steps.  One bench "step" = one complete chain from the naive initialisation.  With
--gpus N every rank samples its own tile (weak scaling) and the detections are all-gathered.

Prints ONE JSON line (rank 0).  `value` = proposals/s over the whole job with the score maps
already resident in HBM.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
F32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X FP32 matrix (= vector) peak, same guide; the U-Nets run in float32 like the reference's
UNET_FLOP_PER_PIXEL = 568896.0   # PosNet 281 472 + ShapeNet 287 424 (torch.utils.flop_counter on the reference modules)
# The roof the chain kernel actually sits under (profiles/r02b_lanes.md): issue of (mostly float64) vector instructions.
# One SIMD issues one wave64 vector instruction per 4 cycles (MI355X_MICROARCH.md, "vector-instruction ISSUE cost"):
N_SIMD, CLOCK_GHZ = 1024, 2.4
VALU_PEAK_GINSTR = N_SIMD * CLOCK_GHZ / 4.0        # 614.4 G wave-instructions / s on 256 CUs


def load_model():
    from mpp_cnn_rs_object_detection_amd import energies
    setup = energies.LegacyEnergySetup()
    setup.load_calibration(os.path.join(REPO, "models_storage", "mpp", "mpp_hrcM"))
    with open(os.path.join(REPO, "model_configs", "mpp", "mpp_hrcM.json")) as f:
        cfg = json.load(f)
    comb = energies.hierarchical_from_manual(cfg["manual"])
    unit, pair = setup.make_energies()
    return setup, energies.build_model_desc(unit, pair, comb)


def bytes_per_proposal(n_points: float, n_cells: int, acc: float) -> float:
    """B = 20*(1+25*rho) + 245 + 20*acc  (state records of the 5x5 cells around the
    proposal + score-map reads + write-back on accept)."""
    rho = n_points / max(n_cells, 1)
    return 20.0 * (1.0 + 25.0 * rho) + 245.0 + 20.0 * acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--iters", type=int, default=100001)
    ap.add_argument("--tile", type=int, default=512)
    ap.add_argument("--objects", type=int, default=200)
    ap.add_argument("--spec", type=int, default=int(os.environ.get("MPP_SPEC_WAVES", "8")))
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("MPP_SPEC_LANES", "0")),
                    help="lane mode: 4 waves x LANES lanes, one speculative step per lane (0 = one wave per step)")
    ap.add_argument("--deep", type=int, default=int(os.environ.get("MPP_DEEP", "128")),
                    help="deep rounds: most steps of one round (csrc/mpp_deep.hip; 0 = one wave per step, the round-2 kernel)")
    ap.add_argument("--dataset-images", type=int, default=56,
                    help="extra measurement, never part of `value` (0 = skip): this many 600x600 images (9 tiles each, score maps given, "
                         "mpp_hrcM) through the batched dataset path, the images dealt to the --gpus ranks (MPPModel.infer's sharding)")
    ap.add_argument("--tiles-per-gpu", type=int, default=1)
    ap.add_argument("--cpu-baseline-chains", type=int, default=36)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-convergence", action="store_true", help="skip the (untimed) wall-clock-to-convergence chain")
    ap.add_argument("--batched-tiles", type=int, default=4096,
                    help="extra measurement, never part of `value` (0 = skip): this many 256x256 tiles (the reference's tile size, "
                         "50 objects, mpp_hrcM schedule of 30257 steps) sampled concurrently, one workgroup per tile")
    ap.add_argument("--batched-spec", type=int, default=1)
    ap.add_argument("--batched-tile", type=int, default=256)
    ap.add_argument("--batched-objects", type=int, default=50)
    ap.add_argument("--batched-capacity", type=int, default=128, help="point slots per chain in the batched run")
    ap.add_argument("--scene", type=int, default=4096,
                    help="extra measurement, never part of `value` (0 = skip): BASELINE config 5 end to end -- a SCENE x SCENE image of "
                         "the reference's image recipe (~5 000 rectangles at 4096) through PosNet + ShapeNet + epilogues + one chain "
                         "per 256-px tile + merge + scores, the tiles (and the nets' regions) dealt to the --gpus ranks")
    ap.add_argument("--scene-dtype", default="float32", choices=["float32", "bfloat16"])
    ap.add_argument("--mosaic", type=int, default=4,
                    help="extra measurement, never part of `value` (0 = skip): BASELINE config 4 -- a MOSAIC x MOSAIC mosaic of the 512-px "
                         "tile generator (2048 x 2048 at 4), score maps given, mpp_log, through MPPModel.infer_image, tiles dealt to the ranks")
    args = ap.parse_args()

    import torch
    from mpp_cnn_rs_object_detection_amd import distributed as mdist
    from mpp_cnn_rs_object_detection_amd import hip_api, kernels, mappings, synth

    # RCCL ("nccl") by default; MPP_DIST_BACKEND=gloo lets several ranks share one GPU for a functional rehearsal
    rank, world = mdist.init_process_group(backend=os.environ.get("MPP_DIST_BACKEND"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the sampler has no CPU fallback")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    gather_device = device if torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl" else None

    setup, model = load_model()
    maps = mappings.default_mappings()
    T = args.tiles_per_gpu
    tiles = [synth.make_tile(args.tile, args.objects, tile_id=rank * T + i) for i in range(T)]
    ctx = hip_api.MppContext(local, point_capacity=1024, spec_waves=args.spec, spec_lanes=args.lanes, deep=args.deep)
    ctx.set_maps(np.stack([t.det for t in tiles]), [np.stack([t.marks[k] for t in tiles]) for k in range(3)])
    ctx.set_model(model, maps)
    ctx.naive_init(setup.detection_threshold, 6.0)
    inits = [ctx.get_points(i) for i in range(T)]
    intensity = np.array([max(1, len(xy)) for xy, _ in inits], dtype=np.float64)
    kd = kernels.make_kernels(maps, 1.0)
    ctx.set_kernels(kd, intensity=intensity)
    T0, alpha, Tt = 1.0, 0.999, 0.0
    n_cells = int(np.ceil(args.tile / 32)) ** 2

    def one_chain(seed):
        for i, (xy, mk) in enumerate(inits):
            ctx.set_points(i, xy, mk)
        ctx.set_schedule(T0, alpha, Tt)
        ctx.run(args.iters, seed=seed, chain0=rank * T)
        if world > 1:          # the product's exchange: records packed on the device, ONE all-gather (RCCL over xGMI)
            ctx.pack_detections(tile_ids, anchors0, gather_cap, gather_buf)
            mdist.all_gather_detections(gather_buf, device=gather_device)
        return ctx.last_kernel_ms()

    tile_ids, anchors0 = np.arange(rank * T, rank * T + T), np.zeros((T, 2), np.int32)
    gather_cap = mdist.gather_capacity(world * T, world)
    gather_buf = torch.zeros((gather_cap + 1, mdist.RECORD), dtype=torch.float64, device=device) if world > 1 else None

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # acceptance rate and mean population for the algorithmic-bytes figure: one traced chain (untimed)
    for i, (xy, mk) in enumerate(inits):
        ctx.set_points(i, xy, mk)
    ctx.set_schedule(T0, alpha, Tt)
    tr_out, _ = ctx.run(args.iters, seed=0, chain0=rank * T, trace_tile=0)
    acc = float(tr_out["accepted"].mean())
    mean_n = float(tr_out["n_after"].mean())

    if world > 1:
        # communicator set-up (RCCL ring over xGMI) is not part of a step: one throw-away gather before anything is timed,
        # also when the driver asks for --warmup 0
        ctx.pack_detections(tile_ids, anchors0, gather_cap, gather_buf)
        mdist.all_gather_detections(gather_buf, device=gather_device)
    for w in range(args.warmup):
        one_chain(seed=w)
    barrier()
    t0 = time.perf_counter()
    kms = []
    for s in range(args.steps):
        kms.append(one_chain(seed=s))
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=gather_device or "cpu")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())

    deep_stats = ctx.deep_stats() if args.deep and args.lanes == 0 else None
    proposals = world * T * args.steps * args.iters
    value = proposals / elapsed
    kernel_ms = float(np.mean(kms))
    bpp = bytes_per_proposal(mean_n, n_cells, acc)
    achieved = bpp * T * args.iters / (kernel_ms * 1e-3) / 1e9
    final_xy, _ = ctx.get_points(0)
    d = np.sqrt(((final_xy[:, None, :] - tiles[0].gt_xy[None]) ** 2).sum(-1)) if len(final_xy) else np.zeros((0, 1))
    matched = int((d.min(axis=0) <= 2).sum()) if len(final_xy) else 0

    lanes = {}
    lj = os.path.join(REPO, "profiles", "latest_lanes.json")
    if os.path.exists(lj):           # committed rocprofv3 counters of the same two launches (profiles/tools/pmc_lanes.sh)
        with open(lj) as f:
            lanes = json.load(f)

    def valu_roofline(kind, proposals_per_s, simds_in_use):
        """float64 vector-instruction issue: achieved = measured instructions per proposal (committed PMC pass) x the
        proposal rate measured live; peak = one wave instruction per SIMD per 4 cycles on all 1024 SIMDs"""
        L = lanes.get(kind)
        if not L:
            return None
        ach = L["valu_instructions_per_proposal"] * proposals_per_s / 1e9
        return {"bound": "fp64-valu-issue", "achieved": ach, "peak": VALU_PEAK_GINSTR, "unit": "G wave-instructions/s",
                "frac": ach / VALU_PEAK_GINSTR,
                "frac_of_the_simds_in_use": ach / (VALU_PEAK_GINSTR * simds_in_use / N_SIMD), "simds_in_use": simds_in_use,
                "valu_instructions_per_proposal": L["valu_instructions_per_proposal"],
                "salu_instructions_per_proposal": L["salu_instructions_per_proposal"],
                "exec_lanes_per_valu_cycle": L["exec_lanes_per_valu_cycle"],
                "useful_lanes_note": ("deep rounds: every active lane works on a DIFFERENT step (draw, densities, unit terms, Green "
                                      "ratio) or a different (step, neighbour) pair; `exec_lanes_per_valu_cycle` therefore counts "
                                      "distinct work, not the wave-uniform repetition it counted for the round-2 kernel (45 of 64)"
                                      if "mpp_deep" in str(L.get("kernel", "")) else
                                      "EXEC is wide because wave-uniform work -- Philox, the draw, densities, the Green ratio -- runs "
                                      "redundantly in every lane; lanes doing DISTINCT work: one per candidate neighbour in eval_delta, "
                                      "4-8 in the clipper, 1 elsewhere"),
                "counters_source": lanes.get("source"), "counter_kernel": L["kernel"]}

    traffic, traffic_src, valu_busy = None, None, None
    tj = os.path.join(REPO, "profiles", "latest_traffic.json")
    if os.path.exists(tj):          # PMC passes cannot run inside this timed process; this is the committed rocprofv3
        with open(tj) as f:         # measurement of the same command (profiles/run_profile.sh), per launch
            t = json.load(f)
        if t.get("iters_per_launch") == args.iters and T == 1:
            traffic, traffic_src = t["hbm_bytes_per_launch"], t["source"]
            valu_busy = t.get("valu_busy_frac_per_simd_of_the_occupied_cu")

    result = {
        "metric": "MPP proposals/s per 512x512 tile",
        "value": value, "unit": "proposals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"{T} x single {args.tile}x{args.tile} synthetic tile per GPU, {args.objects} objects, "
                        f"mpp_hrcM energies, {args.iters} RJMCMC steps per chain (T0=1, alpha=0.999), naive init",
            "iters_per_step": args.iters, "tiles_per_gpu": T, "spec_waves": args.spec, "spec_lanes": args.lanes,
            "parallelism": f"tile-parallel x{world}, one all-gather of detections" if world > 1 else "1 tile, 1 workgroup",
            "accept_rate": acc, "mean_points": mean_n, "final_points": int(len(final_xy)),
            "gt_matched_within_2px": matched, "gt_objects": int(len(tiles[0].gt_xy)),
        },
    }
    # the sampler is priced against HBM bandwidth with the algorithmic bytes per proposal -- that is
    # `roofline.frac`.  The kernel does not live under that roof (the configuration is in LDS, measured traffic is below the
    # algorithmic bytes; one chain occupies 1 of 256 CUs and is bound by the latency of its own dependency chain): what it
    # does with the instruction issue slots of the SIMDs it occupies is reported beside it as `roofline.issue`.
    issue = valu_roofline("one", T * args.iters / (kernel_ms * 1e-3), 4 * T) if T == 1 else None
    roof = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_source": traffic_src, "traffic_unit": "bytes per launch",
        "algorithmic_bytes_per_launch": bpp * T * args.iters, "algorithmic_bytes_per_proposal": bpp,
        "kernel": "mpp_deep_kernel" if deep_stats else "mpp_chain_kernel", "kernel_ms": kernel_ms,
        "kernel_note": ("a step of this bench = one chain of `iters` proposals = TWO launches since the hot start: mpp_chain_kernel (one "
                        "wave per step) for the first proposals, until ~5 of 8 commit per round, then mpp_deep_kernel for the rest "
                        f"({deep_stats['committed'] // max(1, T)} of {args.iters} here); `kernel_ms` = HIP-event time of "
                        "both, which is what the rocprofv3 summary's two averages add up to") if deep_stats and deep_stats.get("committed", 0) and
                       deep_stats["committed"] < T * args.iters else None,
        "issue": issue, "occupied_cu_valu_busy_frac": valu_busy,
        "note": "one chain = one workgroup (8 waves) = 1 of 256 CUs: `frac` prices it against the whole chip's HBM roof as SURVEY "
                "8(d) asks; `issue.frac_of_the_simds_in_use` is the share of instruction issue slots of the 4 SIMDs it runs on; "
                "many chains: see `batched.roofline`",
    }
    if deep_stats and deep_stats["rounds"]:
        roof["deep_rounds"] = {
            "rounds_per_launch": deep_stats["rounds"] / T, "steps_evaluated_per_committed": deep_stats["evaluated"] / max(1, deep_stats["committed"]),
            "steps_committed_per_round": deep_stats["committed"] / deep_stats["rounds"],
            "rounds_with_a_second_pass": deep_stats["rounds_with_change"],
            "useful_lanes_per_wave_in_the_lane_phases": deep_stats["evaluated"] / deep_stats["rounds"] / args.spec,
            "note": "every active lane evaluates a DIFFERENT step (no wave-uniform work repeated in 64 lanes as in the round-2 kernel): "
                    "`useful_lanes...` of 64 lanes are active while proposals are drawn; the neighbour evaluation spreads the wave's "
                    "(step, neighbour) pairs over all lanes",
        }
    result["roofline"] = roof

    if rank == 0 and T >= 1 and not args.no_convergence:
        # Wall-clock to convergence (BASELINE metric), untimed extra chain of tile 0 run in chunks:
        # (i) the whole schedule, (ii) time-to-quality = first chunk end with >= 95 % of the synthetic centres matched
        # within 2 px and the energy within 1 % of its final value.  Chunked launches continue the same Philox chain.
        chunk, done_steps, cum_ms, hist = 2000, 0, 0.0, []
        for i, (xy, mk) in enumerate(inits):
            ctx.set_points(i, xy, mk)
        ctx.set_schedule(T0, alpha, Tt)
        gt = tiles[0].gt_xy
        while done_steps < args.iters:
            n_run = min(chunk, args.iters - done_steps)
            ctx.run(n_run, seed=0, chain0=rank * T)
            cum_ms += ctx.last_kernel_ms()
            done_steps += n_run
            pxy, _ = ctx.get_points(0)
            dd = np.sqrt(((pxy[:, None, :] - gt[None]) ** 2).sum(-1)) if len(pxy) else np.full((1, len(gt)), 1e9)
            hist.append((done_steps, cum_ms, float((dd.min(axis=0) <= 2).mean()), float(ctx.total_energy(0))))
        e_final = hist[-1][3]
        ttq = next(((st, ms) for st, ms, m, e in hist if m >= 0.95 and abs(e - e_final) <= 0.01 * abs(e_final)), None)
        result["config"]["convergence"] = {
            "schedule_steps": args.iters, "schedule_wall_s": kernel_ms * 1e-3,
            "time_to_quality_steps": ttq[0] if ttq else None,
            "time_to_quality_s": ttq[1] * 1e-3 if ttq else None,
            "criterion": ">= 95 % of the centres within 2 px and energy within 1 % of the final energy, checked every 2000 steps",
            "final_energy": e_final, "final_matched_fraction": hist[-1][2],
            "reference_probe": "reference NumPy sampler: 41.4 s for 30 257 steps of this tile (145/200 matched), BASELINE.md section 2",
        }

    if args.scene > 0:
        # BASELINE config 5 (strong scaling over --gpus: ONE image, its tiles dealt to the ranks): every rank generates the
        # same image and the same seeded random-init nets, runs the nets on ITS region + halo, samples its tiles, takes part
        # in the gather / score all-reduces.  Wrapped: a failure here must not cost the headline line.
        try:
            from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
            from mpp_cnn_rs_object_detection_amd.mpp_model import MPPModel
            from mpp_cnn_rs_object_detection_amd.shapes import Rectangle
            S = args.scene
            img, sc_xy, _ = synth.make_scene_image((S, S), int(5250 * (S / 4096) ** 2), noise=0.02, seed=5)
            nets = synth.random_score_nets(0, local, getattr(torch, args.scene_dtype))
            synth.calibrate_div_clf(nets, img[:min(S, 1024), :min(S, 1024)])
            with open(os.path.join(REPO, "model_configs", "mpp", "mpp_hrcM.json")) as f:
                cfg = json.load(f)
            cwd = os.getcwd()
            os.chdir(REPO)
            try:
                mpp = MPPModel(cfg, phase="val", load=True, nets=nets, device=local)
            finally:
                os.chdir(cwd)
            data = ImageWMaps(name="0005", shape=(S, S), image=img, detection_map=None, param_dist_maps=None,
                              mappings=maps, param_names=Rectangle.PARAMETERS, gt_config=[])
            best = None
            for rep in range(3):                              # (the first pass also pays MIOpen's algorithm search)
                mpp.rng = np.random.default_rng(0)
                barrier()
                t0 = time.perf_counter()
                region = mpp.region_maps(data, rank, world)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                pts, scores = mpp.infer_image(data, rank, world, region_data=region)
                barrier()
                t2 = time.perf_counter()
                tt = torch.tensor([t1 - t0, t2 - t0], dtype=torch.float64, device=gather_device or "cpu")
                if world > 1:
                    torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
                cur = {"nets_s": float(tt[0]), "total_s": float(tt[1]), "chain_kernel_ms_rank0": mpp.last_run["kernel_ms"],
                       "detections": len(pts), "tiles": len(mpp.last_run["anchors"]), "tiles_rank0": len(mpp.last_run["mine"]),
                       "steps_per_chain": mpp.last_run["total_steps"], "region_rank0": [int(v) for v in region.shape] if region is not None else None}
                if best is None or cur["total_s"] < best["total_s"]:
                    best = cur
            best["nets_tflops"] = UNET_FLOP_PER_PIXEL * S * S / best["nets_s"] / 1e12 if world == 1 and best["nets_s"] > 0 else None
            best["nets_frac_of_f32_matrix_peak"] = (best["nets_tflops"] / F32_MATRIX_PEAK_TFLOPS
                                                    if best["nets_tflops"] and args.scene_dtype == "float32" else None)
            best["sample_merge_score_s"] = best["total_s"] - best["nets_s"]
            result["scene"] = dict(best, image=S, rectangles_in_image=int(len(sc_xy)), nets_dtype=args.scene_dtype, ranks=world,
                                   proposals_per_s=best["tiles"] * best["steps_per_chain"] / best["total_s"],
                                   note="BASELINE config 5 end to end (max over ranks; best of 3): image recipe of data/make_synth_data.py:16-47, "
                                        "seeded random-init nets (no trained model.pt here), mpp_hrcM; with --gpus N the tiles of this ONE image "
                                        "are dealt to the ranks (strong scaling), each rank's nets cover its own region + halo only")
            del mpp, nets, region
            torch.cuda.empty_cache()
        except Exception as e:                                   # noqa: BLE001
            result["scene"] = {"error": f"{type(e).__name__}: {e}"}

    if args.mosaic > 0:
        # BASELINE config 4: the mosaic with the learned-weights config (no-calibration energies + logistic combinator)
        try:
            from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
            from mpp_cnn_rs_object_detection_amd.mpp_model import MPPModel
            from mpp_cnn_rs_object_detection_amd.shapes import Rectangle
            mdet, mmarks, mgt, _ = synth.make_mosaic(args.mosaic, 512, 200)
            with open(os.path.join(REPO, "model_configs", "mpp", "config_mpp_log.json")) as f:
                cfg = json.load(f)
            cwd = os.getcwd()
            os.chdir(REPO)
            try:
                mpp = MPPModel(cfg, phase="val", load=True, device=local)
            finally:
                os.chdir(cwd)
            S = 512 * args.mosaic
            data = ImageWMaps(name="0004", shape=(S, S), image=None, detection_map=mdet, param_dist_maps=mmarks, mappings=maps,
                              param_names=Rectangle.PARAMETERS, gt_config=[])
            best = None
            for rep in range(2):
                mpp.rng = np.random.default_rng(0)
                barrier()
                t0 = time.perf_counter()
                pts, scores = mpp.infer_image(data, rank, world)
                barrier()
                tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=gather_device or "cpu")
                if world > 1:
                    torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
                c_xy = np.array([[q.x, q.y] for q in pts], dtype=float).reshape(-1, 2)
                from scipy.spatial import cKDTree
                found = float((cKDTree(c_xy).query(mgt.astype(float))[0] <= 2).mean()) if len(c_xy) else 0.0
                cur = {"total_s": float(tt[0]), "chain_kernel_ms_rank0": mpp.last_run["kernel_ms"], "detections": len(pts),
                       "objects": int(len(mgt)), "matched_within_2px": found, "tiles": len(mpp.last_run["anchors"]),
                       "tiles_rank0": len(mpp.last_run["mine"]), "steps_per_chain": mpp.last_run["total_steps"]}
                if best is None or cur["total_s"] < best["total_s"]:
                    best = cur
            result["mosaic"] = dict(best, image=S, ranks=world, config="mpp_log",
                                    proposals_per_s=best["tiles"] * best["steps_per_chain"] / best["total_s"],
                                    note="BASELINE config 4 end to end from host score maps (upload of the rank's region, naive init, one "
                                         "launch, gather, merge, scores; max over ranks, best of 2); tiles dealt to the --gpus ranks")
            del mpp
        except Exception as e:                                   # noqa: BLE001
            result["mosaic"] = {"error": f"{type(e).__name__}: {e}"}

    if args.dataset_images > 0:
        # Dataset inference (the reference's outer loop, mpp_model.py:220-262, is serial over images): the images are dealt to
        # the ranks in blocks -- what MPPModel.infer does under torchrun -- and every rank runs its block through the batched
        # path (tiles of many images in one launch).  Strong scaling over --gpus: the same images, more ranks.
        try:
            from mpp_cnn_rs_object_detection_amd.custom_types import ImageWMaps
            from mpp_cnn_rs_object_detection_amd.mpp_model import MPPModel
            from mpp_cnn_rs_object_detection_amd.shapes import Rectangle
            with open(os.path.join(REPO, "model_configs", "mpp", "mpp_hrcM.json")) as f:
                cfg = json.load(f)
            cwd = os.getcwd()
            os.chdir(REPO)
            try:
                mpp = MPPModel(cfg, phase="val", load=True, device=local)
            finally:
                os.chdir(cwd)
            n_img = args.dataset_images
            mine = mdist.shard_tiles(n_img, rank, world)
            images = []
            for k in mine:
                gt_xy, gt_marks = synth.make_gt(600, 260, tile_id=700 + k)
                ddet, dmarks = synth.render_maps((600, 600), gt_xy, gt_marks)
                images.append(ImageWMaps(name=f"{k:04}", shape=(600, 600), image=None, detection_map=ddet, param_dist_maps=dmarks,
                                         mappings=maps, param_names=Rectangle.PARAMETERS, gt_config=[]))
            seeds_all = [int(v) for v in np.random.default_rng(0).integers(0, 2 ** 63 - 1, size=n_img)]
            best = None
            for rep in range(2):
                barrier()
                t0 = time.perf_counter()
                n_det = 0
                # 28 images x 9 tiles = 252 chains per launch; as in MPPModel.infer a worker thread brings the next batch's score
                # maps to the GPU (side stream) while this batch's chains run
                from concurrent.futures import ThreadPoolExecutor
                side = torch.cuda.Stream(device=device)

                def upload(b):
                    with torch.cuda.stream(side):
                        r = mpp._regions_of_batch(images[b:b + 28])
                    side.synchronize()
                    return r

                with ThreadPoolExecutor(max_workers=1) as pool:
                    fut = pool.submit(upload, 0) if images else None
                    for b in range(0, len(images), 28):
                        regions = fut.result()
                        fut = pool.submit(upload, b + 28) if b + 28 < len(images) else None
                        res = mpp.infer_images(images[b:b + 28], regions, image_seeds=[seeds_all[k] for k in mine[b:b + 28]])
                        n_det += sum(len(r[0]) for r in res)
                barrier()
                tt = torch.tensor([time.perf_counter() - t0, float(n_det)], dtype=torch.float64, device=gather_device or "cpu")
                if world > 1:
                    t_max = tt[:1].clone()
                    torch.distributed.all_reduce(t_max, op=torch.distributed.ReduceOp.MAX)
                    torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.SUM)
                    tt[0] = t_max[0]
                if best is None or float(tt[0]) < best["total_s"]:
                    best = {"total_s": float(tt[0]), "detections": int(tt[1])}
            result["dataset"] = dict(best, images=n_img, images_rank0=len(mine), tiles_per_image=9, steps_per_chain=30257, ranks=world,
                                     images_per_s=n_img / best["total_s"], sharding="by image, contiguous blocks (MPPModel.infer)",
                                     note="600x600 images with host score maps (upload, naive init, chains, merge, scores; max over ranks, "
                                          "best of 2); with --gpus N the SAME images are dealt to N ranks (strong scaling)")
            del mpp
        except Exception as e:                                   # noqa: BLE001
            result["dataset"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0 and world == 1 and args.batched_tiles > 0:
        B, bt, bobj, biters = args.batched_tiles, args.batched_tile, args.batched_objects, 30257
        base = [synth.make_tile(bt, bobj, tile_id=1000 + i) for i in range(min(B, 8))]
        reps = (B + len(base) - 1) // len(base)
        B = reps * len(base)                       # 8 distinct tiles, `reps` chains (own chain id) on each of them
        bctx = hip_api.MppContext(local, point_capacity=args.batched_capacity, spec_waves=args.batched_spec, replicas=reps, deep=args.deep)
        bctx.set_maps(np.stack([t.det for t in base]), [np.stack([t.marks[k] for t in base]) for k in range(3)])
        bctx.set_model(model, maps)
        bctx.naive_init(setup.detection_threshold, 6.0)
        binten = np.maximum(1, bctx.counts()[:B]).astype(np.float64)
        bctx.set_kernels(kernels.make_kernels(maps, 1.0), intensity=binten)
        bctx.set_schedule(T0, alpha, Tt)
        bctx.run(2000, seed=1)                                        # warm-up
        bctx.naive_init(setup.detection_threshold, 6.0)
        bctx.set_schedule(T0, alpha, Tt)
        tb = time.perf_counter()
        bctx.run(biters, seed=2)
        wall = time.perf_counter() - tb
        kms = bctx.last_kernel_ms()
        n_end = bctx.counts()[:B]
        brate = B * biters / (kms * 1e-3)
        bbpp = bytes_per_proposal(float(n_end.mean()), (bt // 32) ** 2, acc)
        result["batched"] = {
            "tiles": B, "distinct_tiles": len(base), "point_capacity": args.batched_capacity, "tile": bt, "objects": bobj, "iters": biters, "spec_waves": args.batched_spec,
            "proposals_per_s": brate, "kernel_ms": kms, "wall_s": wall, "mean_final_points": float(n_end.mean()),
            "roofline": {"bound": "hbm", "achieved": bbpp * brate / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": bbpp * brate / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_proposal": bbpp,
                         "issue": valu_roofline("many", brate, N_SIMD)},
            "deep_rounds": bctx.deep_stats() if args.deep else None,
            "note": "one workgroup per tile, all tiles in one launch; the reference's own parallel axis",
        }
        bctx.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle                     # the CPU restatement, timed as the baseline ("port")
        from concurrent.futures import ThreadPoolExecutor
        kd_cpu = kernels.make_kernels(maps, float(intensity[0]))

        def cpu_chains(seeds):
            """`len(seeds)` complete chains of the bench tile on one thread (the C call releases the GIL)"""
            o = oracle.Oracle(tiles[0].shape, tiles[0].det, tiles[0].marks, model, kd_cpu)
            for s in seeds:
                o.set_points(*inits[0])
                o.set_temperature(T0, alpha, Tt)
                o.run(args.iters, s, chain=0)
            return len(seeds)

        t1 = time.perf_counter()
        cpu_chains(range(args.cpu_baseline_chains))
        dt = time.perf_counter() - t1
        result["cpu_baseline"] = {
            "value": args.cpu_baseline_chains * args.iters / dt, "unit": "proposals/s", "cores": 1, "kind": "port",
            "sample": f"{args.cpu_baseline_chains} chains x {args.iters} steps of the same tile "
                      f"(oracle/mpp_oracle.c, gcc -O2, 1 thread, {os.cpu_count()} host cores present)",
            "reference_python_probe": "0.73e3 proposals/s (reference NumPy sampler, 512x512/200 objects, 1 core of "
                                      "the build container, BASELINE.md section 2; the reference cannot travel to the GPU box)",
        }
        # all host cores this job may use: the reference's own parallel mode is a process pool over tiles
        # (train_utils.py:11-18); here one chain of the same tile per worker thread, each with its own oracle state.  The box
        # may show more cores than the job's CPU quota lets run at once, so the pool is timed at a few sizes up to every
        # visible core and the BEST rate is the baseline (all of them are reported).
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        avail = max(1, min(avail, int(os.environ.get("MPP_CPU_CORES", str(avail)))))
        sizes = sorted({min(avail, 16), min(avail, 64), avail})
        sweep = []
        for cores in sizes:
            per = 2 if cores > 32 else 4
            t1 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=cores) as pool:
                done_chains = sum(pool.map(cpu_chains, [range(1000 * w, 1000 * w + per) for w in range(cores)]))
            dt = time.perf_counter() - t1
            sweep.append({"threads": cores, "chains_per_thread": per, "proposals_per_s": done_chains * args.iters / dt})
        bestc = max(sweep, key=lambda d: d["proposals_per_s"])
        result["cpu_baseline_all_cores"] = {
            "value": bestc["proposals_per_s"], "unit": "proposals/s", "cores": bestc["threads"], "kind": "port",
            "sample": f"{bestc['threads']} threads x {bestc['chains_per_thread']} chains x {args.iters} steps, one tile per thread (the "
                      f"reference's Pool-over-tiles mode, train_utils.py:11-18); {avail} cores visible to the job, {os.cpu_count()} on the host; "
                      f"best of the pool sizes tried",
            "pool_sizes_tried": sweep,
            "gpu_over_cpu_one_tile": value / bestc["proposals_per_s"] if T == 1 else None,
            "gpu_batched_over_cpu": (result.get("batched", {}).get("proposals_per_s") or 0.0) / bestc["proposals_per_s"] or None,
        }
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
