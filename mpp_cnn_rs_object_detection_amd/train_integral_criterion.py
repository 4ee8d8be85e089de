"""Learning the energy weights with the integral criterion: lower the energy of the valid configurations, raise the
energy of sampled invalid ones, ``loss = E(x+)/n+ - E(x-)/n- + reg`` (reference
``train_energy_combination/train_integral_criterion.py:20-258``; the deprecated ``grad_descent`` config key selects it
too, ``mpp_model.py:142-154``).

Negatives come from the sampler itself run with the current weights from the ground truth (``'rjmcmc'``, all images of
a batch in ONE launch), from the NumPy perturbation presets (``'perturbation'``) or from kernel walks (``'kernel'``);
the per-point energy vectors of every configuration come from the from-scratch kernel behind ``EPointsSet``
(``compute_many_energy_vectors`` of ``energies/energy_utils.py``).
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch
from torch.optim.lr_scheduler import ExponentialLR

from .custom_types import ImageWMaps
from .perturbation_sampler import sample_multiple_kernel_perturbations, sample_perturbations
from .point_set import EPointsSet
from .sampler import sample_rjmcmc_batch
from .weight_models import init_model


def compute_many_energy_vectors(configurations, image_config: ImageWMaps, ue, pe, energy_names, device: int = 0,
                                _ctx=None) -> np.ndarray:
    """[sum of the configurations' sizes][n_terms], columns in ``energy_names`` order"""
    rows, ctx = [], _ctx
    for cfg in configurations:
        pts = EPointsSet(list(cfg), image_config.shape[:2], ue, pe, image_data=image_config, device=device, _ctx=ctx)
        ctx = pts._ctx
        if len(pts) == 0:
            continue
        vec = pts.energy_vectors()
        rows.append(np.stack([np.asarray(vec[n], dtype=np.float64) for n in energy_names], axis=-1))
    return np.concatenate(rows, axis=0) if rows else np.zeros((0, len(energy_names)))


def _sample(method: str, image_data: List[ImageWMaps], rng, samples_per_image: int, energy_setup, weights_model, device,
            rjmcmc_params=None, pert_config=None):
    if method == "single":
        return [[d.gt_config] for d in image_data], 1
    if method == "rjmcmc":
        comb = weights_model.get_energy_combination_function()
        shapes = {tuple(d.shape[:2]) for d in image_data}
        groups = [image_data] if len(shapes) == 1 else [[d] for d in image_data]
        out = []
        for g in groups:
            out += sample_rjmcmc_batch(g, rng=rng, num_samples=samples_per_image, energy_combinator=comb, init_config="gt",
                                       energy_setup=energy_setup, device=device, **rjmcmc_params)
        return out, samples_per_image
    if method == "perturbation":
        return [sample_perturbations(image_data=d, rng=rng, n_samples=samples_per_image, **pert_config)
                for d in image_data], samples_per_image
    if method == "kernel":
        res = [sample_multiple_kernel_perturbations(d, n_samples=samples_per_image, rng=rng, energy_setup=energy_setup,
                                                    device=device, **pert_config) for d in image_data]
        return [[list(s) for s in r] for r in res], samples_per_image
    raise ValueError(method)


def train_integral_criterion(train_loader, rng: np.random.Generator, logger, energy_setup, samples_per_image: int,
                             n_epochs: int, save_dir: str, neg_sampling_method: str = "rjmcmc",
                             pos_sampling_method: str = "single", reg_weight=None, optim: str = "adam",
                             lr_scheduler: bool = False, learning_rate=1e-1, weight_model_type="hierarchical",
                             multiprocess=True, device: int = 0, **kwargs):
    energy_names = energy_setup.energy_names
    weights_model = init_model(weight_model_type, energy_setup=energy_setup, **kwargs)
    print(f"initial weights: {weights_model.as_dict()}")
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
            optimiser.zero_grad()
            neg, n_neg = _sample(neg_sampling_method, image_data, rng, samples_per_image, energy_setup, weights_model, device,
                                 kwargs.get("rjmcmc_params"), kwargs.get("neg_pert_config"))
            pos, n_pos = _sample(pos_sampling_method, image_data, rng, samples_per_image, energy_setup, weights_model, device,
                                 kwargs.get("rjmcmc_params"), kwargs.get("pos_pert_config"))
            v_rows, nv_rows = [], []
            for i, d in enumerate(image_data):
                ue, pe = energy_setup.make_energies(image_data=d)
                v_rows.append(compute_many_energy_vectors(pos[i], d, ue, pe, energy_names, device))
                nv_rows.append(compute_many_energy_vectors(neg[i], d, ue, pe, energy_names, device))
            x_plus = torch.tensor(np.concatenate(v_rows, axis=0), dtype=torch.float32)
            x_minus = torch.tensor(np.concatenate(nv_rows, axis=0), dtype=torch.float32)
            e_plus = torch.div(weights_model.forward(x_plus), n_pos)
            e_minus = torch.div(weights_model.forward(x_minus), n_neg)
            reg = (reg_weight * weights_model.regularisation_term(E_plus=e_plus, E_minus=e_minus)
                   if reg_weight is not None and reg_weight != 0.0 else 0)
            loss = e_plus - e_minus + reg
            loss.backward()
            optimiser.step()
            log = {"batch": batch_id, "loss": float(loss.detach()), "e_plus": float(e_plus.detach()),
                   "n_e_plus": len(x_plus) / n_pos, "e_minus": float(e_minus.detach()), "n_e_minus": len(x_minus) / n_neg,
                   "reg": float(reg.detach()) if torch.is_tensor(reg) else float(reg),
                   "lr": learning_rate if scheduler is None else scheduler.get_last_lr()[0], **weights_model.as_dict()}
            logger.update(epoch=epoch_id, metrics=log)
            print(f"[epoch {epoch_id + 1}/{n_epochs}][batch {batch_id + 1}/{n_batches}] loss: {log['loss']:.4f} "
                  f"e_plus: {log['e_plus']:.4f} e_minus: {log['e_minus']:.4f} lr {log['lr']:.4f} reg {log['reg']:.4f}")
        if scheduler is not None:
            scheduler.step()
        print(weights_model.as_dict())
    return weights_model.get_energy_combination_function()
