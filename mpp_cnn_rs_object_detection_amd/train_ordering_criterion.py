"""Learning the energy weights with the ordering criterion: the ground truth must have a lower energy than its
perturbations, ``loss = -mean_k [E(perturbed_k) - E(gt)]`` (reference
``train_energy_combination/train_ordering_criterion.py:43-219``).

What runs where: the negatives are kernel walks of the chain kernel with every proposal applied
(``perturbation_sampler.sample_multiple_kernel_perturbations``); the energy vectors of the points each aggregated
perturbation touches come from ONE launch of ``mpp_delta_vectors`` per image; torch differentiates the small
weight model on those rows.  The reference evaluates one ``energy_delta`` at a time through the Python energy graph;
the criterion, its sign conventions and the ``delta != 0`` filter are the same.
"""
from __future__ import annotations

import json
import logging
import os
import time
from datetime import datetime
from typing import Dict, List, Sequence

import numpy as np
import torch
from torch.optim.lr_scheduler import ExponentialLR

from .custom_types import ImageWMaps, Perturbation
from .perturbation_sampler import sample_multiple_kernel_perturbations
from .point_set import EPointsSet
from .weight_models import init_model


class Logger:
    """The part of ``utils/logger.py:14-62`` the criterion uses: lists per key, rewritten to ``log.json``."""

    def __init__(self, save_dir: str):
        self.log: Dict[str, list] = {}
        self.save_dir = save_dir

    def update(self, epoch: int, metrics: Dict[str, float], prefix: str = ""):
        row = {"epoch": epoch, "timestamp": datetime.now().strftime("%m/%d/%y-%H:%M:%S"),
               **{prefix + k: v for k, v in metrics.items()}}
        for k, v in row.items():
            self.log.setdefault(k, []).append(v)
        if self.save_dir:
            with open(os.path.join(self.save_dir, "log.json"), "w") as f:
                json.dump(self.log, f, indent=1, default=float)


def perturbation_rows(points: EPointsSet, perturbations: Sequence[Perturbation], names: Sequence[str] = None):
    """rows [R][n_terms], sign [R] (+1: energy after, -1: energy before), case [R] for a list of perturbations of one
    configuration: exactly the points whose energy vector ``EnergyGraph.compute_delta`` (energy_graph.py:139-225)
    evaluates on both sides, minus those whose vector does not change (their two terms cancel bit for bit)."""
    before, after, mask = points.energy_delta_vectors(perturbations, names=names)   # columns in the weight model's order
    rows, sign, case = [], [], []
    for k in range(len(perturbations)):
        m = mask[k]
        both = np.where(m == 1)[0]
        both = both[np.any(before[k, both] != after[k, both], axis=1)]
        b_idx = np.concatenate([both, np.where(m == 2)[0]])
        a_idx = np.concatenate([both, np.where(m == 3)[0]])
        rows += [before[k, b_idx], after[k, a_idx]]
        sign += [-np.ones(len(b_idx)), np.ones(len(a_idx))]
        case += [np.full(len(b_idx) + len(a_idx), k)]
    nt = before.shape[-1]
    return (np.concatenate(rows).reshape(-1, nt) if rows else np.zeros((0, nt)),
            np.concatenate(sign) if sign else np.zeros(0), np.concatenate(case).astype(np.int64) if case else np.zeros(0, np.int64))


def criterion_loss(weights_model, per_image_rows, device="cpu"):
    """-mean of the non-zero energy deltas (train_ordering_criterion.py:101-118); None when there is none"""
    deltas = []
    for rows, sign, case, n_cases in per_image_rows:
        if len(rows) == 0:
            continue
        x = torch.tensor(rows, dtype=torch.float32, device=device)        # EnergyComputeTorch._dict_to_tensor: .float()
        e = weights_model.point_energies(x) * torch.tensor(sign, dtype=torch.float32, device=device)
        d = torch.zeros(n_cases, dtype=torch.float32, device=device).index_add(0, torch.tensor(case, device=device), e)
        deltas.append(d[d.detach() != 0.0])
    if not deltas or sum(len(d) for d in deltas) == 0:
        return None
    return -torch.mean(torch.cat(deltas))


def train_ordering_criterion(train_loader, rng: np.random.Generator, logger: Logger, samples_per_image: int,
                             n_epochs: int, save_dir: str, energy_setup, neg_sampling_method: str = "rjmcmc",
                             pos_sampling_method: str = "single", reg_weight=None, optim: str = "adam",
                             lr_scheduler: bool = False, learning_rate=1e-1, weight_model_type="hierarchical",
                             multiprocess=True, device: int = 0, **kwargs):
    weights_model = init_model(weight_model_type, energy_setup=energy_setup, **kwargs)
    if optim == "adam":
        optimiser = torch.optim.Adam(params=weights_model.parameters(), lr=learning_rate)
    elif optim == "sgd":
        optimiser = torch.optim.SGD(params=weights_model.parameters(), lr=learning_rate)
    else:
        raise ValueError(optim)
    scheduler = ExponentialLR(optimiser, **kwargs["lr_scheduler_params"]) if lr_scheduler else None
    n_batches = len(train_loader)
    for epoch_id in range(n_epochs):
        for batch_id, image_data in enumerate(train_loader):
            image_data: List[ImageWMaps]
            optimiser.zero_grad()
            start = time.perf_counter()
            per_image = []
            n_pert = 0
            for d in image_data:
                uec, pec = energy_setup.make_energies(image_data=d)
                d.gt_config_set = EPointsSet(points=d.gt_config, support_shape=d.shape[:2], unit_energies_constructors=uec,
                                             pair_energies_constructors=pec, image_data=d, device=device)
                if len(d.gt_config) == 0:
                    continue
                perts = sample_multiple_kernel_perturbations(d, energy_setup=energy_setup, rng=rng,
                                                             **kwargs["neg_pert_config"], n_samples=samples_per_image,
                                                             return_perturbations=True, aggregate_pert=True, device=device)
                rows, sign, case = perturbation_rows(d.gt_config_set, perts, names=energy_setup.energy_names)
                per_image.append((rows, sign, case, len(perts)))
                n_pert += len(perts)
            logging.info(f"sampling and evaluating {n_pert} perturbations in {time.perf_counter() - start:.2f}s")
            loss = criterion_loss(weights_model, per_image)
            if loss is None:
                continue
            loss.backward()
            optimiser.step()
            log = {"batch": batch_id, "loss": float(loss.detach()),
                   "lr": learning_rate if scheduler is None else scheduler.get_last_lr()[0], **weights_model.as_dict()}
            logger.update(epoch=epoch_id, metrics=log)
            print(f"[epoch {epoch_id + 1}/{n_epochs}][batch {batch_id + 1}/{n_batches}] loss: {log['loss']:.4f} "
                  f"lr {log['lr']:.4f} ")
        if scheduler is not None:
            scheduler.step()
        print(weights_model.as_dict())
    return weights_model.get_energy_combination_function()
