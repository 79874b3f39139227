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

    # ---- differentiable slots (torch expressions of the Parameters, used for gradients only) ----
    def _diff_geom(self):
        return None

    def _diff_slots_surface(self):
        return {"Rj": self.transform.rot, "tj": self.transform.trans, "geom": self._diff_geom()}

    def _diff_slots(self, j: int):
        return self._diff_slots_surface()

    # ---- reference API ------------------------------------------------------------------------
    def intersectTest(self, rays):
        return K.intersect_test(rays, self._surface_record()[None])

    def forward(self, rays, *args):
        """(t, hit_global, normal_global, hit_local).  When a parameter of this surface requires grad,
        `hit_global` is differentiable with respect to it (the other outputs are returned detached)."""
        rec = self._surface_record()
        o = K.surface_forward(rays, rec[None], 0, want_phys=False)
        hit_global = o["hit_global"]
        from .. import autograd as AG
        if AG.needs_grad([self]):
            # hit_global = position after a sequential "trace" of this one surface with Transmit physics
            rec["phys_kind"] = P.PHYS_TRANSMIT
            outs = AG.trace_sequential_with_grad([_AsElement(self)], rays, rec[None])
            moved = outs[0].t()
            hit_global = torch.where((o["t"] < float("inf"))[:, None], moved, hit_global)
        return o["t"], hit_global, o["normal_global"], o["hit_local"]

    @property
    def z(self):
        return self.transform.trans[2]

    def __len__(self):
        return 1

    @property
    def surfaces(self):
        return (self,)


class _AsElement:
    """Adapter: lets a bare Surface stand where the gradient code expects an element."""

    def __init__(self, surface):
        self.shape = surface

    def parameters(self):
        return self.shape.parameters()

    def modules(self):
        return self.shape.modules()

    def _diff_slots(self, j):
        d = self.shape._diff_slots(j)
        d["phys"] = None
        return d


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

    def _diff_geom(self):
        z = torch.zeros_like(self.radius)
        return torch.stack([self.radius, self.radius * self.radius, z, z])


class Cylinder(Surface):
    """x^2 + y^2 = R^2 along local z (geom/primitives.py:190-241)."""
    _surf_kind = P.SURF_CYLINDER

    def __init__(self, radius: float, transform: Optional[RayTransform] = None, radius_grad: bool = False):
        super().__init__(transform)
        self.radius = nn.Parameter(torch.as_tensor(radius, dtype=torch.float32), requires_grad=radius_grad)

    def _fill_geometry(self, rec):
        r = P.scalar(self.radius)
        rec["geom"][:2] = (r, r * r)

    def _diff_geom(self):
        z = torch.zeros_like(self.radius)
        return torch.stack([self.radius, self.radius * self.radius, z, z])


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

    def _diff_geom(self):
        return torch.stack([self.c, self.k, self.c * (1 + self.k), torch.zeros_like(self.c)])


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

    def _diff_geom(self):
        z = torch.zeros_like(self.slope)
        return torch.stack([self.slope, self.slope * self.slope, z, z])


class WedgeYZ(Cone):
    """Declared and refused in the reference as well (geom/primitives.py:500-503)."""

    def __init__(self):
        raise NotImplementedError()
