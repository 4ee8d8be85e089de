"""Differentiable weight models of the energy combination (torch), trained by ``train_ordering_criterion``.

Mirrors the reference's ``energies/combination/{base,hierarchical,logistic}.py``: same parameters and initial
values, same ``forward`` (sum of the per-point energies of the rows it is given), same ``as_dict`` keys (they are
the columns of ``log.json``), and ``get_energy_combination_function()`` returns the NumPy combinator the sampler
flattens into its term table.  ``point_energies`` is the per-row form the batched criterion needs.
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch
from torch import Tensor, nn
from torch.nn import Module, functional

from .energies import HierarchicalEnergyCombinator, LogisticEnergyCombinator, sigmoid


class WeightModel:
    """``combination/base.py:8-27``"""

    def point_energies(self, x: Tensor) -> Tensor:
        raise NotImplementedError

    def forward(self, x: Tensor):
        return torch.sum(self.point_energies(x))

    def as_dict(self):
        raise NotImplementedError

    def get_energy_combination_function(self):
        raise NotImplementedError

    def get_decision_function(self):
        raise NotImplementedError

    def regularisation_term(self, **kwargs):
        raise NotImplementedError


class LogisticEnergyModel(Module, WeightModel):
    """``combination/logistic.py:29-72``: E_u = 2 sigmoid(sum_k (b + w_k v_k)) - 1"""

    def __init__(self, energy_names: List[str], use_bias: bool = True):
        super().__init__()
        self.energy_names = list(energy_names)
        self.weights = nn.Parameter(torch.tensor([1.0] * len(self.energy_names)))
        if use_bias:
            self.bias = nn.Parameter(torch.tensor(0.0))
        else:
            self.register_buffer("bias", torch.tensor(0.0))

    def point_energies(self, x: Tensor) -> Tensor:
        return 2 * torch.sigmoid(torch.sum(self.bias + self.weights * x, dim=-1)) - 1

    def forward(self, x: Tensor):
        return torch.sum(self.point_energies(x))

    def get_np_weights(self):
        return self.weights.detach().cpu().numpy(), float(self.bias.detach().cpu())

    def as_dict(self):
        w, b = self.get_np_weights()
        return {**{k + "_weight": float(w[i]) for i, k in enumerate(self.energy_names)}, "bias": b}

    def get_energy_combination_function(self):
        w, b = self.get_np_weights()
        return LogisticEnergyCombinator(weights=w, bias=b, energy_names=self.energy_names)

    def get_decision_function(self):
        w, b = self.get_np_weights()
        return lambda vector: 2 * sigmoid(np.sum(b + w * vector, axis=-1)) - 1

    def regularisation_term(self, **kwargs):
        return torch.tensor(0.0)


class HierarchicalEnergyModel(Module, WeightModel):
    """``combination/hierarchical.py:51-108`` (legacy setup: Position, Shape | Overlap, Alignment, Area)"""

    def __init__(self, threshold: float, learn_bias: bool = False):
        super().__init__()
        self.data_prior_weight = nn.Parameter(torch.tensor([1.0, 1.0]))
        self.data_weight = nn.Parameter(torch.tensor([1.0, 1.0]))
        self.prior_weight = nn.Parameter(torch.tensor([1.0, 1.0, 1.0]))
        self.threshold_detection = threshold
        if learn_bias:
            self.bias = nn.Parameter(torch.tensor(0.0))
        else:
            self.register_buffer("bias", torch.tensor(0.0))

    def _softmaxed(self):
        return (functional.softmax(self.data_prior_weight, dim=0), functional.softmax(self.data_weight, dim=0),
                functional.softmax(self.prior_weight, dim=0))

    def point_energies(self, x: Tensor) -> Tensor:
        dp, dw, pw = self._softmaxed()
        indicator = torch.less_equal(x[:, 0], self.threshold_detection)
        data_term = dw[0] * x[:, 0] + indicator * dw[1] * x[:, 1]
        prior_term = indicator * (pw[0] * x[:, 2] + pw[1] * x[:, 3] + pw[2] * x[:, 4])
        return dp[0] * data_term + dp[1] * prior_term + self.bias

    def forward(self, x: Tensor):
        return torch.sum(self.point_energies(x))

    def regularisation_term(self, **kwargs):
        dp, dw, pw = self._softmaxed()
        return sum(torch.square(1 - v) for v in (dp[0], dp[1], dw[0], dw[1], pw[0], pw[1], pw[2]))

    def _np(self):
        dp, dw, pw = (t.detach().cpu().numpy() for t in self._softmaxed())
        return dp, dw, pw, float(self.threshold_detection), float(self.bias.detach().cpu())

    def get_energy_combination_function(self):
        dp, dw, pw, thr, b = self._np()
        return HierarchicalEnergyCombinator(weights_data=dw, weights_prior=pw, data_prior_weights=dp,
                                            detection_threshold=thr, bias=b)

    def get_decision_function(self):
        dp, dw, pw, thr, b = self._np()

        def fun(vector: np.ndarray):
            ind = np.less_equal(vector[:, 0], thr)
            data = dw[0] * vector[:, 0] + ind * dw[1] * vector[:, 1]
            prior = ind * (pw[0] * vector[:, 2] + pw[1] * vector[:, 3] + pw[2] * vector[:, 4])
            return dp[0] * data + dp[1] * prior + b

        return fun

    def as_dict(self):
        dp, dw, pw, thr, b = self._np()
        return {"data_weight": float(dp[0]), "prior_weight": float(dp[1]), "PositionEnergy_indicator_threshold": thr,
                "PositionEnergy_data_weight": float(dw[0]), "ShapeEnergy_data_weight": float(dw[1]),
                "RectangleOverlapEnergy_prior_weight": float(pw[0]), "ShapeAlignmentEnergy_prior_weight": float(pw[1]),
                "AreaPriorEnergy_prior_weight": float(pw[2]), "bias": b}


def init_model(weight_model_type: str, energy_setup, **kwargs):
    """``train_energy_combination/train_utils.py:21-41`` (the mlp / linear / loghrc variants are not part of the
    shipped configs and are not built)"""
    if weight_model_type == "hierarchical":
        return HierarchicalEnergyModel(threshold=0.0, **kwargs.get("weights_model_params", {}))
    if weight_model_type == "logistic":
        return LogisticEnergyModel(use_bias=True, energy_names=energy_setup.energy_names)
    raise ValueError(f"weight_model_type {weight_model_type!r} is not built (hierarchical, logistic)")
