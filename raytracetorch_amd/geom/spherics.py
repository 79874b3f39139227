"""Rotationally symmetric lens stacks: spherical caps plus cylindrical edges
(reference: geom/spherics.py:10-298).

Surface order is optical faces first (front to back), then one edge cylinder per glass element.
Shape-level validity (geom/spherics.py:27-46), in the element frame:
  face  j <  N_optical : x^2 + y^2 <= radius^2
  edge  j >= N_optical : sag_e(radius) <= z <= sag_{e+1}(radius),  e = j - N_optical
"""
from typing import Optional, Sequence

import torch
import torch.nn as nn

from .. import program as P
from .bounded import HalfSphere
from .primitives import Cylinder
from .shape import Shape
from .transform import RayTransform


class Spheric(Shape):

    def _make_surface(self, C, z_vertex, c_grad=False, z_grad=False):
        t = RayTransform(translation=[0., 0., z_vertex], trans_grad=z_grad, trans_mask=[False, False, True])
        return HalfSphere(curvature=C, transform=t, curvature_grad=c_grad)

    def _build(self, curvatures: Sequence[float], thicknesses: Sequence[float], D: float,
               c_grads: Sequence[bool], z_grad: bool, D_grad: bool):
        """Centre the stack on z = 0, create faces then edges, validate."""
        self.N_optical = len(curvatures)
        self.radius = nn.Parameter(torch.as_tensor(D / 2.0, dtype=torch.float32), requires_grad=D_grad)
        z = -sum(thicknesses) / 2.0
        faces = []
        for i, (C, g) in enumerate(zip(curvatures, c_grads)):
            faces.append(self._make_surface(C, z, c_grad=g, z_grad=z_grad))
            if i < len(thicknesses):
                z = z + thicknesses[i]
        edges = []
        for _ in thicknesses:
            e = Cylinder(D / 2)
            e.radius = self.radius  # shared Parameter (geom/spherics.py:92-93)
            edges.append(e)
        for s in faces + edges:
            self.surfaces.append(s)
        with torch.no_grad():
            for i, C in enumerate(curvatures):
                if abs(0.5 * C) > 1 / D:
                    raise ValueError(f"|R{i + 1}| must be larger than D/2")
            for i, T in enumerate(thicknesses):
                if T <= 1e-6:
                    raise ValueError("Thickness T must be positive" if len(thicknesses) == 1
                                     else f"Thickness T{i + 1} must be positive")
            sags = [f.sagittalZ(self.radius) for f in faces]
            for i in range(len(sags) - 1):
                if sags[i] > sags[i + 1]:
                    raise ValueError("Intersecting optical surfaces" if len(sags) == 2
                                     else f"Optical surfaces {i + 1} and {i + 2} intersect")

    def _fill_shape_bound(self, rec, j: int):
        r = P.scalar(self.radius)
        if j < self.N_optical:
            rec["shape_kind"] = P.SHP_SPHERIC_FACE
            rec["shp"][0] = r * r
        else:
            e = j - self.N_optical
            lo, hi = self.surfaces[e], self.surfaces[e + 1]
            rec["shape_kind"] = P.SHP_SPHERIC_EDGE
            rec["shp"][0] = P.sag_f32(P.scalar(lo.c), P.host_value(lo.transform.trans)[2], r)
            rec["shp"][1] = P.sag_f32(P.scalar(hi.c), P.host_value(hi.transform.trans)[2], r)

    def inBounds(self, local_pos, surf_idx):
        if surf_idx >= self.N_optical:
            e = surf_idx - self.N_optical
            z = local_pos[:, 2]
            return (z >= self.surfaces[e].sagittalZ(self.radius)) & (z <= self.surfaces[e + 1].sagittalZ(self.radius))
        return (local_pos[:, 0] ** 2 + local_pos[:, 1] ** 2) <= self.radius ** 2

    def _thickness(self, i):
        return self.surfaces[i + 1].transform.trans[2] - self.surfaces[i].transform.trans[2]

    @property
    def T(self):
        return self.surfaces[self.N_optical - 1].transform.trans[2] - self.surfaces[0].transform.trans[2]

    @property
    def T_edge(self):
        return self.surfaces[self.N_optical - 1].sagittalZ(self.radius) - self.surfaces[0].sagittalZ(self.radius)


class Singlet(Spheric):
    """[front cap @ -T/2, back cap @ +T/2, edge cylinder] (geom/spherics.py:56-111)."""

    def __init__(self, C1: float, C2: float, D: float, T: float,
                 C1_grad: bool = True, C2_grad: bool = True, D_grad: bool = False, T_grad: bool = True,
                 transform: Optional[RayTransform] = None):
        super().__init__(transform=transform)
        self._build([C1, C2], [T], D, [C1_grad, C2_grad], T_grad, D_grad)


class Doublet(Spheric):
    """Three caps, two edges (geom/spherics.py:116-206)."""

    def __init__(self, C1: float, C2: float, C3: float, D: float, T1: float, T2: float,
                 C1_grad: bool = True, C2_grad: bool = True, C3_grad: bool = True, D_grad: bool = False,
                 T1_grad: bool = True, T2_grad: bool = True, transform: Optional[RayTransform] = None):
        super().__init__(transform=transform)
        self._build([C1, C2, C3], [T1, T2], D, [C1_grad, C2_grad, C3_grad], T1_grad or T2_grad, D_grad)

    T1 = property(lambda self: self._thickness(0))
    T2 = property(lambda self: self._thickness(1))


class Triplet(Spheric):
    """Four caps, three edges (geom/spherics.py:209-298)."""

    def __init__(self, C1: float, C2: float, C3: float, C4: float, D: float,
                 T1: float, T2: float, T3: float,
                 C1_grad: bool = True, C2_grad: bool = True, C3_grad: bool = True, C4_grad: bool = True,
                 D_grad: bool = False, T1_grad: bool = True, T2_grad: bool = True, T3_grad: bool = True,
                 transform: Optional[RayTransform] = None):
        super().__init__(transform=transform)
        self._build([C1, C2, C3, C4], [T1, T2, T3], D, [C1_grad, C2_grad, C3_grad, C4_grad],
                    T1_grad or T2_grad or T3_grad, D_grad)

    T1 = property(lambda self: self._thickness(0))
    T2 = property(lambda self: self._thickness(1))
    T3 = property(lambda self: self._thickness(2))
