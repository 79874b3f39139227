"""Paraxial ("ideal") elements: a plane or disk carrying a ``Linear`` functor, and the 5x5 paraxial
matrices (reference: elements/ideal.py:9-163).

The trace of an ideal element runs in the same kernels as every other surface: the ``Linear`` functor
writes physics kind ``RTT_PHYS_LINEAR`` with (Cx, Cy, Dx, Dy) into the surface record and rotates with the
surface's own transform.
"""
from typing import Optional

import torch
import torch.nn as nn

from ..geom import Disk, Plane, RayTransform
from ..phys import Linear
from .parent import Element


def _with(mat, rows, cols, vals):
    return mat.index_put((torch.tensor(rows), torch.tensor(cols)), vals)


def ParaxialLensMat(lens_power_x: torch.Tensor, lens_power_y: torch.Tensor) -> torch.Tensor:
    """elements/ideal.py:9-15"""
    eye = torch.eye(5, device=lens_power_x.device, dtype=lens_power_x.dtype)
    return _with(eye, [1, 3], [0, 2], torch.stack([-lens_power_x, -lens_power_y]))


def ParaxialDistMat(dist: torch.Tensor) -> torch.Tensor:
    """elements/ideal.py:17-23"""
    eye = torch.eye(5, device=dist.device, dtype=dist.dtype)
    return _with(eye, [0, 2], [1, 3], torch.stack([dist, dist]))


def ParaxialRefractMat(Cx, Cy, ior_1, ior_2: torch.Tensor) -> torch.Tensor:
    """elements/ideal.py:25-37"""
    vals = torch.stack([Cx * (ior_1 - ior_2) / ior_2, Cy * (ior_1 - ior_2) / ior_2, ior_1 / ior_2, ior_1 / ior_2])
    eye = torch.eye(5, device=ior_1.device, dtype=ior_1.dtype)
    return _with(eye, [1, 3, 1, 3], [0, 2, 1, 3], vals)


def ParaxialMirrorMat(Cx, Cy: torch.Tensor) -> torch.Tensor:
    """elements/ideal.py:39-45"""
    eye = torch.eye(5, device=Cx.device, dtype=Cx.dtype)
    return _with(eye, [1, 3], [0, 2], torch.stack([-2 * Cx, -2 * Cy]))


def _aperture(diameter, transform):
    return Plane(transform=transform) if diameter == float("inf") else Disk(radius=diameter / 2, transform=transform)


class LinearElement(Element):
    """A planar surface with a Linear functor that shares the surface's transform (elements/ideal.py:47-61)."""

    def __init__(self, shape: Plane, linSurfFunc: Linear):
        super().__init__()
        self.shape = shape
        linSurfFunc.transform = shape.transform
        linSurfFunc._rot_is_surface = True
        self.surface_functions.append(linSurfFunc)

    def _paraxial(self):
        sf = self.surface_functions[0]
        return ParaxialLensMat(-sf.Cx, -sf.Cy)


class IdealThinLens(LinearElement):
    """Cx = Cy = P = -1/focal (elements/ideal.py:64-87)."""

    def __init__(self, focal: float, focal_grad: bool = False, diameter: float = float("inf"),
                 transform: Optional[RayTransform] = None):
        super().__init__(shape=_aperture(diameter, transform), linSurfFunc=Linear())
        self.P = nn.Parameter(torch.as_tensor(-1 / focal, dtype=torch.float32), requires_grad=focal_grad)
        self.surface_functions[0].Cx = self.P
        self.surface_functions[0].Cy = self.P

    @property
    def f(self):
        return -1 / self.P


class IdealCylThinLens(LinearElement):
    """elements/ideal.py:90-118.  The reference assigns Cy on ``surface_functions[1]``, which does not
    exist, so its constructor raises IndexError; the same statement is kept here."""

    def __init__(self, focal_x: float, focal_y: float, focal_x_grad: bool = False, focal_y_grad: bool = False,
                 diameter: float = float("inf"), transform: Optional[RayTransform] = None):
        super().__init__(shape=_aperture(diameter, transform), linSurfFunc=Linear())
        self.Px = nn.Parameter(torch.as_tensor(-1 / focal_x, dtype=torch.float32), requires_grad=focal_x_grad)
        self.Py = nn.Parameter(torch.as_tensor(-1 / focal_y, dtype=torch.float32), requires_grad=focal_y_grad)
        self.surface_functions[0].Cx = self.Px
        self.surface_functions[1].Cy = self.Py

    @property
    def fx(self):
        return -1 / self.Px

    @property
    def fy(self):
        return -1 / self.Py


class IdealMirror(LinearElement):
    """Px = -2/Rx, Py = -2/Ry on a Linear functor (elements/ideal.py:121-163).  Like the reference's, the
    functor keeps Dx = Dy = 1, so the outgoing ray continues towards +z of the mirror frame."""

    def __init__(self, radius_x: float, radius_y: float, radius_x_grad: bool = False, radius_y_grad: bool = False,
                 diameter: float = float("inf"), transform: Optional[RayTransform] = None):
        super().__init__(shape=_aperture(diameter, transform), linSurfFunc=Linear())
        self.Px = nn.Parameter(torch.as_tensor(-2.0 / radius_x, dtype=torch.float32), requires_grad=radius_x_grad)
        self.Py = nn.Parameter(torch.as_tensor(-2.0 / radius_y, dtype=torch.float32), requires_grad=radius_y_grad)
        self.surface_functions[0].Cx = self.Px
        self.surface_functions[0].Cy = self.Py

    def _paraxial(self):
        return ParaxialMirrorMat(-self.Px / 2, -self.Py / 2)

    @property
    def fx(self):
        return -1 / self.Px

    @property
    def fy(self):
        return -1 / self.Py

    @property
    def Rx(self):
        return -2 / self.Px

    @property
    def Ry(self):
        return -2 / self.Py
