"""RJMCMC sampling driver: ``sample_rjmcmc`` and its batched, GPU-native form.

Mirrors the reference's ``models/mpp/rjmcmc_sampler/sample_rjmcmc.py:23-102`` (same signature,
same schedule arithmetic, same "which state is returned" rule) while running the chain of every
tile in ONE kernel launch, one workgroup per tile (``libmppgpu.so``: ``mpp_run``).

Differences a caller can observe (both documented in DESIGN.md):

* the proposal stream is Philox4x32-10 keyed by a seed drawn from the given ``rng`` (the reference
  consumes the NumPy generator directly and is not reproducible run to run);
* with ``use_split_merge`` the kernel needs ``spec_waves`` 1 or 8 (split / merge steps run alone on the live state).
"""
from __future__ import annotations

import logging
import time
from typing import List, Optional, Sequence, Union

import numpy as np

from . import energies as E
from .custom_types import ImageWMaps
from .hip_api import MppContext
from .kernels import make_kernels
from .shapes import Rectangle

NMS_DISTANCE = 6.0      # reference sample_rjmcmc.py:27


def _to_rectangles(xy: np.ndarray, marks: np.ndarray) -> List[Rectangle]:
    # (.tolist() first: building 5 000 rectangles from NumPy scalars took 54 ms of a 0.26 s scene, from Python numbers 8 ms)
    return [Rectangle(x, y, size=s, ratio=r, angle=a) for (x, y), (s, r, a) in zip(xy.tolist(), marks.tolist())]


def resolve_schedule(num_samples: int, init_temperature: float, alpha_t, burn_in, samples_interval,
                     target_temperature: float, iter_multiplier=None):
    """The reference's schedule arithmetic (``sample_rjmcmc.py:58-66,78`` and ``stopping.py:41-42``).

    Returns (alpha, T_target, total_steps, snapshot_step): the chain runs ``max_iter + 1`` steps and the
    configuration handed back is the one after the last step t with t >= burn_in and
    t % samples_interval == 0 (``rjmcmc.py:138-141``), or the final one if no such step exists."""
    if iter_multiplier is not None:
        burn_in = burn_in * iter_multiplier
        samples_interval = samples_interval * iter_multiplier
        alpha_t = np.power(alpha_t, 1 / iter_multiplier)
    if isinstance(alpha_t, str):
        if alpha_t != "auto":
            raise ValueError("alpha_t must be a number or 'auto'")
        alpha_t = float(np.power(target_temperature / init_temperature, 1 / burn_in))
        target_temperature = 0
    burn_in, samples_interval = int(burn_in), int(samples_interval)
    max_iter = burn_in + (num_samples + 1) * samples_interval
    total = max_iter + 1
    snaps = [t for t in range(burn_in, total) if samples_interval > 0 and t % samples_interval == 0]
    return float(alpha_t), float(target_temperature), total, snaps


#: relative speed of ONE chain with 8 / 4 / 2 / 1 waves in deep rounds (csrc/mpp_deep.hip; the 512-px bench tile: 117 / 131 /
#: 177 / 218 ms per 100 001 steps, gpurun_out of round 3; one wave per step, round 1: 1.0 / 0.74 / 0.47 / 0.30)
SPEC_SPEED = {8: 1.0, 4: 0.89, 2: 0.66, 1: 0.54}
#: ... and with one wave per step (the kernels chains with the split / merge kernels run on): profiles/r01_batched_sweep.json
SPEC_SPEED_WAVE = {8: 1.0, 4: 0.74, 2: 0.47, 1: 0.30}
LDS_PER_CU, WAVES_PER_CU, N_CU = 160 * 1024, 8, 256      # MI355X; 8 waves of 256 VGPRs fill a CU


def choose_spec_waves(ctx: MppContext, n_tiles: int, use_split_merge: bool = False) -> int:
    """Speculative waves per chain for a launch of ``n_tiles`` chains: fewer waves per chain let more chains share a
    CU -- if their LDS footprint (set by the point capacity) allows it.  Minimises rounds-of-workgroups / chain speed.
    With the default capacity of 1024 slots one chain fills a CU's LDS and 8 waves are always best; with 128 slots the
    measured optimum is 8 waves up to 256 tiles, 4 up to 512, 2 up to 1024, 1 beyond (same sweep)."""
    best, best_t = 8, None
    for spec in ((8, 1) if use_split_merge else (8, 4, 2, 1)):
        ctx.set_option("spec_waves", spec)
        lds = max(1, ctx.get_option("lds_bytes"))
        resident = N_CU * max(1, min(LDS_PER_CU // lds, WAVES_PER_CU // spec))
        t = -(-n_tiles // resident) / (SPEC_SPEED_WAVE if use_split_merge else SPEC_SPEED)[spec]
        if best_t is None or t < best_t - 1e-12:
            best, best_t = spec, t
    return best


class TileBatchSampler:
    """All tiles of an image (or of a batch of images) sampled concurrently on one GPU."""

    def __init__(self, tiles: Sequence[ImageWMaps], energy_setup, energy_combinator, device: int = 0,
                 point_capacity: int = 1024, spec_waves: Optional[int] = 8, ctx: Optional[MppContext] = None,
                 use_split_merge: bool = False, keys=None, stacked_maps=None):
        """``keys`` = (seeds, chain ids), one per tile: the Philox key and chain id each tile's chain uses instead of the
        launch's seed and ``chain0 + tile`` -- tiles of several images in one launch keep the chains they would run in a
        launch of their own image (``mpp_set_chain_keys``)."""
        self.use_split_merge = use_split_merge
        shapes = {tuple(t.shape[:2]) for t in tiles}
        if len(shapes) != 1:
            raise ValueError(f"all tiles of a batch must have the same shape, got {shapes}")
        self.tiles = list(tiles)
        self.energy_setup = energy_setup
        self.energy_combinator = energy_combinator
        self.mappings = tiles[0].mappings
        unit, pair = energy_setup.make_energies(tiles[0])
        self.model_units = unit
        self.model = E.build_model_desc(unit, pair, energy_combinator)
        auto_spec = spec_waves is None
        self.ctx = ctx or MppContext(device, point_capacity=point_capacity, spec_waves=8 if auto_spec else spec_waves)
        det0 = tiles[0].detection_map
        if stacked_maps is not None:                  # (det [T,p,p], marks 3 x [T,p,p,32]) already stacked on the GPU
            det, marks = stacked_maps
            if int(det.shape[0]) != len(tiles):
                raise ValueError("stacked_maps: one slice per tile")
        elif hasattr(det0, "data_ptr"):                 # maps already on the GPU (U-Net epilogue output)
            import torch
            det = torch.stack([t.detection_map for t in tiles]).contiguous()
            marks = [torch.stack([t.param_dist_maps[k] for t in tiles]).contiguous() for k in range(3)]
        else:
            det = np.stack([np.asarray(t.detection_map, dtype=np.float32) for t in tiles])
            marks = [np.stack([np.asarray(t.param_dist_maps[k], dtype=np.float32) for t in tiles]) for k in range(3)]
        self.ctx.set_maps(det, marks)
        if E.classic_image(unit) is not None:
            # a classic image energy (contrast setup): every tile brings its own picture, prepared as the setup prescribes
            self.ctx.set_image(np.stack([E.classic_image(energy_setup.make_energies(t)[0]) for t in tiles]))
            if spec_waves not in (1, 8):
                auto_spec = False
                self.ctx.set_option("spec_waves", 8)
        if keys is not None:
            self.ctx.set_chain_keys(keys[0], keys[1])
        self.ctx.set_model(self.model, self.mappings)
        if auto_spec and ctx is None:
            self.ctx.set_option("spec_waves", choose_spec_waves(self.ctx, len(self.tiles), use_split_merge))

    def init(self, init_config: Union[str, None, Sequence[Sequence[Rectangle]]]):
        n = len(self.tiles)
        if isinstance(init_config, str) and init_config == "naive":
            self.ctx.naive_init(self.energy_setup.detection_threshold, NMS_DISTANCE)
        else:
            if isinstance(init_config, str) and init_config == "gt":
                configs = [t.gt_config for t in self.tiles]
            elif init_config is None:
                configs = [[] for _ in range(n)]
            else:
                configs = list(init_config)
                if configs and isinstance(configs[0], Rectangle):
                    configs = [configs] * n
            for i, cfg in enumerate(configs):
                xy = np.array([[p.x, p.y] for p in cfg], dtype=np.int32).reshape(-1, 2)
                mk = np.array([[p.size, p.ratio, p.angle] for p in cfg], dtype=np.float64).reshape(-1, 3)
                self.ctx.set_points(i, xy, mk)
        counts = self.ctx.counts()[:n].astype(np.float64)
        self.intensity = np.maximum(1.0, counts)             # reference sample_rjmcmc.py:68
        self.ctx.set_kernels(make_kernels(self.mappings, 1.0, use_split_merge=self.use_split_merge), intensity=self.intensity)

    def run(self, total_steps: int, snapshot_steps: Sequence[int], num_samples: int, T0: float, alpha: float,
            T_target: float, seed: int, chain0: int = 0, on_device=None, as_arrays: bool = False):
        """-> per tile, the list of the last ``num_samples`` sampled configurations.
        ``on_device``: a callable ``f(ctx)`` that takes each sampled state where it lies (e.g.
        ``ctx.pack_detections`` into an all-gather buffer) instead of copying it to host rectangles.
        ``as_arrays``: configurations as (xy [n, 2] int32, marks [n, 3] float64) instead of lists of ``Rectangle``."""
        self.ctx.set_schedule(T0, alpha, T_target)
        wanted = list(snapshot_steps)[-num_samples:] if snapshot_steps else []
        samples = [[] for _ in self.tiles]

        def take():
            if on_device is not None:
                on_device(self.ctx)
                return
            for i, pts in enumerate(self.ctx.get_points_all()[:len(self.tiles)]):
                samples[i].append(pts if as_arrays else _to_rectangles(*pts))

        done = 0
        self.kernel_ms = 0.0
        for t in wanted:                                       # the state after step t = after t+1 steps
            self._run_resumable(t + 1 - done, seed, chain0)
            done = t + 1
            take()
        if done < total_steps:                                 # the reference keeps stepping to max_iter
            self._run_resumable(total_steps - done, seed, chain0)
        if not wanted:
            take()
        return samples

    def _run_resumable(self, n_steps: int, seed: int, chain0: int):
        self.ctx.run(n_steps, seed, chain0)
        self.kernel_ms += self.ctx.last_kernel_ms()


def sample_rjmcmc_batch(tiles: Sequence[ImageWMaps], rng: np.random.Generator, num_samples: int, energy_combinator,
                        init_config, init_temperature: float, alpha_t, burn_in: int, energy_setup,
                        samples_interval: int, target_temperature: float, verbose: int = 0, iter_multiplier=None,
                        use_split_merge: bool = False, device: int = 0, spec_waves: int = 8,
                        point_capacity: int = 1024, chain0: int = 0):
    """``sample_rjmcmc`` for many equally-shaped tiles at once; returns one result list per tile."""
    alpha, T_target, total, snaps = resolve_schedule(num_samples, init_temperature, alpha_t, burn_in, samples_interval,
                                                      target_temperature, iter_multiplier)
    sampler = TileBatchSampler(tiles, energy_setup, energy_combinator, device=device, spec_waves=spec_waves,
                               point_capacity=point_capacity, use_split_merge=use_split_merge)
    sampler.init(init_config)
    seed = int(rng.integers(0, 2 ** 63 - 1))
    start = time.perf_counter()
    samples = sampler.run(total, snaps, num_samples, init_temperature, alpha, T_target, seed, chain0)
    end = time.perf_counter()
    logging.info(f"rjmcmc on {len(tiles)} tile(s) ran in {end - start:.2f}s ({(end - start) / total:.1e}s/iter, "
                 f"kernel {sampler.kernel_ms:.1f} ms) (int. {sampler.intensity.tolist()} | iter {total - 1} | "
                 f"num_samples {num_samples})")
    return samples


def sample_rjmcmc(image_data: ImageWMaps, rng: np.random.Generator, num_samples: int, energy_combinator,
                  init_config, init_temperature: float, alpha_t, burn_in: int, energy_setup, samples_interval: int,
                  target_temperature: float, verbose: int = 0, iter_multiplier: float = None,
                  use_split_merge: bool = False, **gpu_options):
    """Drop-in for the reference's ``sample_rjmcmc`` (one tile).  Returns ``[points]`` for
    ``num_samples == 1`` and the last ``num_samples`` sampled configurations otherwise."""
    return sample_rjmcmc_batch([image_data], rng, num_samples, energy_combinator, init_config, init_temperature,
                               alpha_t, burn_in, energy_setup, samples_interval, target_temperature, verbose,
                               iter_multiplier, use_split_merge, **gpu_options)[0]
