"""Mirrors: a single reflecting surface as the element's shape (reference: elements/mirror.py:14-164)."""
import math
from typing import Optional

import torch
import torch.nn as nn

from ..geom import Quadric, QuadricZY, RayTransform
from ..geom.bounded import BoundedHalfSphere, HalfCyl
from ..phys import Reflect
from .parent import Element


class Mirror(Element):

    def __init__(self):
        super().__init__()
        self.surface_functions.append(Reflect())

    def _set_d(self, d, d_grad):
        self.d = nn.Parameter(torch.as_tensor(d, dtype=torch.float32), requires_grad=d_grad)

    @property
    def c1(self):
        return self.shape.c

    @property
    def R(self):
        return 1.0 / self.shape.c

    @property
    def f(self):
        return 1.0 / (2.0 * self.shape.c)

    # which of (x, y) the mirror focuses in; the paraxial matrix is that of a sphere of the vertex curvature
    _paraxial_axes = (1.0, 1.0)

    def _paraxial(self):
        """elements/mirror.py:45-52, :79-88, :115-123, :155-164."""
        from .ideal import ParaxialMirrorMat
        c = self.shape.c
        return ParaxialMirrorMat(self._paraxial_axes[0] * c, self._paraxial_axes[1] * c)


class SphericalMirror(Mirror):
    """Spherical cap clipped to `diameter` (elements/mirror.py:22-31)."""

    def __init__(self, c1: float, d: float, diameter: float = float("inf"), c1_grad: bool = False,
                 d_grad: bool = False, diameter_grad: bool = False, transform: Optional[RayTransform] = None):
        super().__init__()
        self.shape = BoundedHalfSphere(curvature=c1, diameter=diameter, curvature_grad=c1_grad,
                                       diameter_grad=diameter_grad, transform=transform)
        self._set_d(d, d_grad)


class CylindricalMirror(Mirror):
    """Cylindrical cap curved along y (elements/mirror.py:55-65)."""

    def __init__(self, c1: float, d: float, c1_grad: bool = False, d_grad: bool = False,
                 transform: Optional[RayTransform] = None):
        super().__init__()
        self.shape = HalfCyl(curvature=c1, curvature_grad=c1_grad, transform=transform)
        self._set_d(d, d_grad)

    _paraxial_axes = (0.0, 1.0)


class ParabolicMirror(Mirror):
    """Paraboloid of revolution, conic constant -1 (elements/mirror.py:91-101)."""

    def __init__(self, c1: float, d: float, c1_grad: bool = False, d_grad: bool = False,
                 transform: Optional[RayTransform] = None):
        super().__init__()
        self.shape = Quadric(c=c1, k=-1.0, c_grad=c1_grad, transform=transform)
        self._set_d(d, d_grad)


class ParabolicMirrorXZ(Mirror):
    """Parabolic trough focusing in x: the y-curved conic turned 90 degrees about z
    (elements/mirror.py:126-141)."""

    def __init__(self, c1: float, d: float, c1_grad: bool = False, d_grad: bool = False,
                 transform: Optional[RayTransform] = None):
        super().__init__()
        translation = transform.trans.detach().tolist() if transform is not None else None
        xz = RayTransform(rotation=[0.0, 0.0, math.pi / 2.0], translation=translation)
        self.shape = QuadricZY(c=c1, k=-1.0, c_grad=c1_grad, transform=xz)
        self._set_d(d, d_grad)

    _paraxial_axes = (1.0, 0.0)


class ParabolicMirrorOffAxis(Mirror):
    """Declared and not implemented in the reference either (elements/mirror.py:167-172)."""

    def __init__(self):
        super().__init__()
        raise NotImplementedError
