"""Surfaces with a finite extent (reference: geom/bounded.py:9-217).

The bound is tested in the surface frame at *both* roots before the minimum is taken
(geom/bounded.py:20-36); the kernels do that from the ``bound_kind`` / ``bnd`` fields filled here.
``inBounds`` stays available as a small tensor predicate because ``ApertureFilter`` is built from
it (elements/aperture.py:17) and user code may call it.
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import program as P
from .primitives import Cone, Plane, Quadric, QuadricZY, Surface, intersectEpsilon
from .transform import RayTransform


class SurfaceBounded(Surface):
    """Mixin: `invert` flips the keep mask (geom/bounded.py:29-30)."""

    def __init__(self, transform: Optional[RayTransform] = None, invert: bool = False, **kw):
        super().__init__(transform=transform, **kw)
        self.invert = invert

    def _fill_bounds(self, rec):
        rec["bound_kind"] = P.BND_ALWAYS

    def _surface_record(self):
        rec = super()._surface_record()
        if getattr(self, "invert", False):
            rec["flags"] |= P.F_INVERT
        return rec

    def inBounds(self, local_pos):
        raise NotImplementedError


class Disk(SurfaceBounded, Plane):
    """x^2 + y^2 <= R^2 on the plane (geom/bounded.py:51-64)."""

    def __init__(self, radius: float, invert: bool = False, transform: Optional[RayTransform] = None):
        super().__init__(transform=transform, invert=invert)
        self.radius = nn.Parameter(torch.as_tensor(radius, dtype=torch.float32), requires_grad=False)

    def _fill_bounds(self, rec):
        r = P.scalar(self.radius)
        rec["bound_kind"] = P.BND_DISK
        rec["bnd"][0] = r * r

    def inBounds(self, local_pos):
        return (local_pos[:, 0] ** 2 + local_pos[:, 1] ** 2) <= self.radius ** 2


class Rectangle(SurfaceBounded, Plane):
    """|x| <= hx and |y| <= hy (geom/bounded.py:67-82)."""

    def __init__(self, half_x: float, half_y: float, invert: bool = False,
                 transform: Optional[RayTransform] = None):
        super().__init__(transform=transform, invert=invert)
        self.hx = nn.Parameter(torch.as_tensor(half_x, dtype=torch.float32))
        self.hy = nn.Parameter(torch.as_tensor(half_y, dtype=torch.float32))

    def _fill_bounds(self, rec):
        rec["bound_kind"] = P.BND_RECT
        rec["bnd"][:2] = (P.scalar(self.hx), P.scalar(self.hy))

    def inBounds(self, local_pos):
        return (local_pos[:, 0].abs() <= self.hx) & (local_pos[:, 1].abs() <= self.hy)


class Ellipse(SurfaceBounded, Plane):
    """Rotated ellipse (geom/bounded.py:85-106)."""

    def __init__(self, r_major: float, r_minor: float, rot: float,
                 r_major_grad: bool = False, r_minor_grad: bool = False, rot_grad: bool = False,
                 invert: bool = False, transform: Optional[RayTransform] = None):
        super().__init__(transform=transform, invert=invert)
        mk = lambda v, g: nn.Parameter(torch.as_tensor(v, dtype=torch.float32), requires_grad=g)
        self.r_minor, self.r_major, self.rot = mk(r_minor, r_minor_grad), mk(r_major, r_major_grad), mk(rot, rot_grad)

    def _fill_bounds(self, rec):
        rot = torch.as_tensor(P.host_value(self.rot))
        rec["bound_kind"] = P.BND_ELLIPSE
        rec["bnd"][:] = (float(torch.cos(rot)), float(torch.sin(rot)),
                         P.scalar(self.r_major), P.scalar(self.r_minor))

    def inBounds(self, local_pos):
        cs, sn = torch.cos(self.rot), torch.sin(self.rot)
        dM = local_pos[:, 0] * cs - local_pos[:, 1] * sn
        dm = local_pos[:, 0] * sn + local_pos[:, 1] * cs
        return ((dM / self.r_major) ** 2 + (dm / self.r_minor) ** 2) <= 1.0


def _sagittal(c, tz, r):
    """Vertex-form sag c r^2 / (1 + sqrt(max(1 - c^2 r^2, 0))) + tz (geom/bounded.py:129-139)."""
    r_sq = r ** 2
    term = torch.relu(1.0 - c ** 2 * r_sq)
    return (c * r_sq) / (1.0 + torch.sqrt(term)) + tz


class HalfSphere(SurfaceBounded, Quadric):
    """Spherical cap: conic with k = 0 kept where |z c| < 1 + 1e-6 (geom/bounded.py:109-139)."""

    def __init__(self, curvature: float, curvature_grad: bool, transform: Optional[RayTransform] = None):
        super().__init__(transform=transform, c=curvature, c_grad=curvature_grad, k=0.0, k_grad=False)

    def _fill_bounds(self, rec):
        rec["bound_kind"] = P.BND_HALF

    def inBounds(self, local_pos):
        return (local_pos[:, 2] * self.c).abs() < 1 + intersectEpsilon

    def sagittalZ(self, radius):
        return _sagittal(self.c, self.transform.trans[2], radius)


class BoundedHalfSphere(HalfSphere):
    """Spherical cap additionally clipped to a circular aperture (geom/bounded.py:142-159)."""

    def __init__(self, curvature: float, diameter: float, curvature_grad: bool = False,
                 diameter_grad: bool = False, transform: Optional[RayTransform] = None):
        super().__init__(curvature=curvature, curvature_grad=curvature_grad, transform=transform)
        self.diameter = nn.Parameter(torch.as_tensor(diameter, dtype=torch.float32), requires_grad=diameter_grad)

    def _fill_bounds(self, rec):
        half = P.scalar(self.diameter) / P.f32(2.0)
        rec["bound_kind"] = P.BND_HALF_DISK
        rec["bnd"][0] = half * half

    def inBounds(self, local_pos):
        xy = local_pos[:, 0] ** 2 + local_pos[:, 1] ** 2
        return super().inBounds(local_pos) & (xy <= (self.diameter / 2.0) ** 2)


class HalfCyl(SurfaceBounded, QuadricZY):
    """Cylindrical cap curved along y (geom/bounded.py:162-186)."""

    def __init__(self, curvature: float, curvature_grad: bool, transform: Optional[RayTransform] = None):
        super().__init__(transform=transform, c=curvature, c_grad=curvature_grad, k=0.0, k_grad=False)

    def _fill_bounds(self, rec):
        rec["bound_kind"] = P.BND_HALF

    def inBounds(self, local_pos):
        return (local_pos[:, 2] * self.c).abs() < 1 + intersectEpsilon

    def sagittalZ(self, y_height):
        return _sagittal(self.c, self.transform.trans[2], y_height)


class SingleCone(SurfaceBounded, Cone):
    """One nappe of the double cone: z * slope >= -1e-6 (geom/bounded.py:189-217)."""

    def __init__(self, slope: float, slope_grad: bool = False, invert: bool = False,
                 transform: Optional[RayTransform] = None):
        # the reference drops `invert` for this class (geom/bounded.py:203-206)
        super().__init__(transform=transform, invert=False, slope=slope, slope_grad=slope_grad)

    def _fill_bounds(self, rec):
        rec["bound_kind"] = P.BND_SINGLE_CONE

    def inBounds(self, local_pos):
        return (local_pos[:, 2] * self.slope) >= -intersectEpsilon
