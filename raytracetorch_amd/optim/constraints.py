"""Log-barrier constraints on the axial layout of a system (reference: optim/constraints.py:10-168).
Host-side functions of the vertex positions (Parameters), differentiable through torch as they stand."""
from typing import List, Optional

import torch
import torch.nn as nn


def _barrier_above(x, lb: float):
    return -torch.log(x - lb)


def _barrier_below(x, ub: float):
    return -torch.log(ub - x)


def _get_optical_z_list(elem):
    """Global vertex z of every optical surface of `elem`: element z plus surface z for the multi-surface lens
    shapes, the element z alone otherwise (optim/constraints.py:29-50)."""
    shape = elem.shape
    base = shape.transform.trans[2]
    if hasattr(shape, "N_optical") and hasattr(shape, "surfaces"):
        return [base + shape.surfaces[i].transform.trans[2] for i in range(shape.N_optical)]
    return [base]


class Constraint(nn.Module):
    pass


def _weighted(weight, terms, elements):
    if not terms:
        return _get_optical_z_list(elements[0])[0].new_zeros(())
    return weight * sum(terms)


class ThicknessConstraint(Constraint):
    """t_min < z_{i+1} - z_i (< t_max) for consecutive optical surfaces inside each element
    (optim/constraints.py:62-106)."""

    def __init__(self, elements: List, t_min: float, t_max: Optional[float] = None, weight: float = 1.0):
        super().__init__()
        self.elements, self.t_min, self.t_max, self.weight = elements, t_min, t_max, weight

    def forward(self):
        terms = []
        for elem in self.elements:
            z = _get_optical_z_list(elem)
            for a, b in zip(z[:-1], z[1:]):
                term = _barrier_above(b - a, self.t_min)
                if self.t_max is not None:
                    term = term + _barrier_below(b - a, self.t_max)
                terms.append(term)
        return _weighted(self.weight, terms, self.elements)


class SpacingConstraint(Constraint):
    """Gap between the last optical surface of an element and the first of the next one > d_min
    (optim/constraints.py:109-140)."""

    def __init__(self, elements: List, d_min: float, weight: float = 1.0):
        super().__init__()
        self.elements, self.d_min, self.weight = elements, d_min, weight

    def forward(self):
        terms = [_barrier_above(_get_optical_z_list(b)[0] - _get_optical_z_list(a)[-1], self.d_min)
                 for a, b in zip(self.elements[:-1], self.elements[1:])]
        return _weighted(self.weight, terms, self.elements)


class SystemLengthConstraint(Constraint):
    """z of the last optical surface minus z of the first < L_max (optim/constraints.py:143-168)."""

    def __init__(self, elements: List, L_max: float, weight: float = 1.0):
        super().__init__()
        self.elements, self.L_max, self.weight = elements, L_max, weight

    def forward(self):
        length = _get_optical_z_list(self.elements[-1])[-1] - _get_optical_z_list(self.elements[0])[0]
        return self.weight * _barrier_below(length, self.L_max)
