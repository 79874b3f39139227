"""Unbounded analytic surfaces (reference: geom/primitives.py:9-497).

A surface object holds its pose and shape parameters and knows how to describe itself as one
``RttSurface`` record.  ``intersectTest`` / ``forward`` keep the reference signatures and run the
per-ray arithmetic (roots of the ray/surface quadratic, epsilon filter, hit points, normals) in the
HIP kernels.
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import kernels as K
from .. import program as P
from .transform import RayTransform

intersectEpsilon = 1e-6


class Surface(nn.Module):
    """Base surface.  Protocol (geom/primitives.py:38-96):
    ``intersectTest(rays) -> t [N,1]``;
    ``forward(rays) -> (t [N], hit_global [N,3], normal_global [N,3], hit_local [N,3])``."""

    _surf_kind = None

    def __init__(self, transform: Optional[RayTransform] = None):
        super().__init__()
        self.epsilon = nn.Parameter(torch.as_tensor(1e-6), requires_grad=False)
        self.transform = RayTransform() if transform is None else transform

    # ---- record description -------------------------------------------------------------------
    def _fill_geometry(self, rec):
        raise NotImplementedError

    def _fill_bounds(self, rec):
        rec["bound_kind"] = P.BND_NONE

    def _surface_record(self):
        """Record of this surface used on its own (the rays' frame is the parent frame)."""
        if self._surf_kind is None:
            raise NotImplementedError(f"{type(self).__name__} has no kernel description")
        rec = P.new_record()
        rec["surf_kind"] = self._surf_kind
        P.set_transform(rec, "j", self.transform)
        self._fill_geometry(rec)
        self._fill_bounds(rec)
        return rec

    def _record(self, j: int):
        """Record when this surface *is* an element's shape (len == 1)."""
        if j != 0:
            raise IndexError("a bare Surface has exactly one surface")
        rec = self._surface_record()
        rec["flags"] |= P.F_SHAPE_ROT_ID
        return rec

    # ---- reference API ------------------------------------------------------------------------
    def intersectTest(self, rays):
        return K.intersect_test(rays, self._surface_record()[None])

    def forward(self, rays, *args):
        o = K.surface_forward(rays, self._surface_record()[None], 0, want_phys=False)
        return o["t"], o["hit_global"], o["normal_global"], o["hit_local"]

    @property
    def z(self):
        return self.transform.trans[2]

    def __len__(self):
        return 1

    @property
    def surfaces(self):
        return (self,)


class Plane(Surface):
    """z = 0 facing +z; t = -oz / dz with |dz| < 1e-6 replaced by 1e-8 (geom/primitives.py:119-143)."""
    _surf_kind = P.SURF_PLANE

    def _fill_geometry(self, rec):
        pass


class Sphere(Surface):
    """|p|^2 = R^2 about the local origin (geom/primitives.py:146-187)."""
    _surf_kind = P.SURF_SPHERE

    def __init__(self, radius: float, radius_grad: bool = False, transform: Optional[RayTransform] = None):
        super().__init__(transform=transform)
        self.radius = nn.Parameter(torch.as_tensor(radius, dtype=torch.float32), requires_grad=radius_grad)

    def _fill_geometry(self, rec):
        r = P.scalar(self.radius)
        rec["geom"][:2] = (r, r * r)


class Cylinder(Surface):
    """x^2 + y^2 = R^2 along local z (geom/primitives.py:190-241)."""
    _surf_kind = P.SURF_CYLINDER

    def __init__(self, radius: float, transform: Optional[RayTransform] = None, radius_grad: bool = False):
        super().__init__(transform)
        self.radius = nn.Parameter(torch.as_tensor(radius, dtype=torch.float32), requires_grad=radius_grad)

    def _fill_geometry(self, rec):
        r = P.scalar(self.radius)
        rec["geom"][:2] = (r, r * r)


class Quadric(Surface):
    """Conic of revolution c(x^2+y^2) + c(1+k)z^2 - 2z = 0 (geom/primitives.py:244-343)."""
    _surf_kind = P.SURF_QUADRIC

    def __init__(self, c: float, k: float, transform: Optional[RayTransform] = None,
                 c_grad: bool = False, k_grad: bool = False):
        super().__init__(transform=transform)
        self.c = nn.Parameter(torch.as_tensor(c, dtype=torch.float32), requires_grad=c_grad)
        self.k = nn.Parameter(torch.as_tensor(k, dtype=torch.float32), requires_grad=k_grad)

    def _fill_geometry(self, rec):
        c, k = P.scalar(self.c), P.scalar(self.k)
        rec["geom"][:3] = (c, k, c * (P.f32(1) + k))


class QuadricZY(Quadric):
    """Cylindrical conic c y^2 + c(1+k) z^2 - 2z = 0, x free (geom/primitives.py:346-395)."""
    _surf_kind = P.SURF_QUADRIC_ZY


class Cone(Surface):
    """Double cone z^2 = s^2 (x^2+y^2) (geom/primitives.py:398-497)."""
    _surf_kind = P.SURF_CONE

    def __init__(self, slope: float, slope_grad: bool = False, transform: Optional[RayTransform] = None):
        super().__init__(transform=transform)
        self.slope = nn.Parameter(torch.as_tensor(slope, dtype=torch.float32), requires_grad=slope_grad)

    def _fill_geometry(self, rec):
        s = P.scalar(self.slope)
        rec["geom"][:2] = (s, s * s)
