"""Multi-surface volumes (reference: geom/shape.py:8-276).

A ``Shape`` owns an element-frame transform and a list of surfaces; a hit on surface ``j`` is kept
only when the hit point, expressed in the element frame, passes ``inBounds(hit, j)``
(geom/shape.py:42-56).  Subclasses describe that test for the kernels through ``_fill_shape_bound``.
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import kernels as K
from .. import program as P
from .primitives import Plane
from .transform import RayTransform


class Shape(nn.Module):

    def __init__(self, transform: Optional[RayTransform] = None):
        super().__init__()
        self.epsilon = nn.Parameter(torch.as_tensor(1e-7), requires_grad=False)
        self.surfaces = nn.ModuleList()
        self.transform = RayTransform() if transform is None else transform

    # ---- record description -------------------------------------------------------------------
    def _fill_shape_bound(self, rec, j: int):
        raise NotImplementedError(f"{type(self).__name__} has no kernel description of inBounds")

    def _record(self, j: int):
        rec = self.surfaces[j]._surface_record()
        rec["flags"] |= P.F_HAS_SHAPE
        P.set_transform(rec, "s", self.transform)
        self._fill_shape_bound(rec, j)
        return rec

    def _diff_slots(self, j: int):
        d = self.surfaces[j]._diff_slots_surface()
        d["Rs"], d["ts"] = self.transform.rot, self.transform.trans
        return d

    def _program(self):
        import numpy as np
        return np.stack([self._record(j) for j in range(len(self.surfaces))])

    # ---- reference API ------------------------------------------------------------------------
    def intersectTest(self, rays):
        """[N, K] distances to every surface, inf where missed or out of bounds (geom/shape.py:25-59)."""
        return K.intersect_test(rays, self._program())

    def forward(self, rays, surf_idx):
        """(t, hit_global, normal_global, hit_local) for one surface, no shape-level bound
        (geom/shape.py:61-87)."""
        o = K.surface_forward(rays, self._record(surf_idx)[None], 0, want_phys=False)
        return o["t"], o["hit_global"], o["normal_global"], o["hit_local"]

    def inBounds(self, local_pos, surf_idx):
        raise NotImplementedError

    @property
    def z(self):
        return self.transform.trans[2]

    def __len__(self):
        return len(self.surfaces)


def _axis_plane(pos, rot_vec, optimize_flag):
    """A canonical plane moved to `pos` with axis-angle `rot_vec`; only the non-zero translation
    components receive gradient (geom/shape.py:239-260)."""
    pos_t = torch.tensor(pos, dtype=torch.float32)
    t = RayTransform(translation=pos_t, rotation=torch.tensor(rot_vec, dtype=torch.float32),
                     trans_grad=optimize_flag, trans_mask=(pos_t.abs() > 1e-5))
    return Plane(transform=t)


class CvxPolyhedron(Shape):
    """Intersection of half-spaces z_q < 1e-4 of its planes (geom/shape.py:104-132).  In a trace the bound
    of surface j is evaluated from the records of the element's other planes, which the scene compiler
    places next to it (shape kind ``SHP_POLYHEDRON``)."""

    def __init__(self, planes_list=None, transform: Optional[RayTransform] = None):
        super().__init__(transform=transform)
        for s in planes_list or ():
            self.surfaces.append(s)

    def _fill_shape_bound(self, rec, j: int):
        rec["shape_kind"] = P.SHP_POLYHEDRON
        rec["surface_id"] = j
        rec["shp"][0] = len(self.surfaces)

    @property
    def PlaneRotMat(self):
        return torch.stack([pl.transform.rot[2, :] for pl in self.surfaces])

    @property
    def PlaneTrans(self):
        return torch.stack([pl.transform.trans for pl in self.surfaces])

    def inBounds(self, local_pos, surf_idx):
        rel = local_pos[None, :, :] - self.PlaneTrans[:, None, :]
        inside = (self.PlaneRotMat[:, None, :] * rel).sum(-1) < 1e-4
        if surf_idx is not None:
            inside[surf_idx, :] = True
        return inside.all(dim=0)


class Box4Side(CvxPolyhedron):
    """Four side planes (+x, -x, +y, -y) of a rectangular prism (geom/shape.py:213-276)."""

    def __init__(self, width: float, height: float, transform: Optional[RayTransform] = None,
                 w_grad: bool = False, h_grad: bool = False):
        super().__init__(transform=transform)
        hp = torch.pi / 2
        self.surfaces.append(_axis_plane([width / 2, 0.0, 0.0], [0.0, -hp, 0.0], w_grad))
        self.surfaces.append(_axis_plane([-width / 2, 0.0, 0.0], [0.0, hp, 0.0], w_grad))
        self.surfaces.append(_axis_plane([0.0, height / 2, 0.0], [hp, 0.0, 0.0], h_grad))
        self.surfaces.append(_axis_plane([0.0, -height / 2, 0.0], [-hp, 0.0, 0.0], h_grad))

    @property
    def width(self):
        return self.surfaces[0].transform.trans[0] - self.surfaces[1].transform.trans[0]

    @property
    def height(self):
        return self.surfaces[2].transform.trans[1] - self.surfaces[3].transform.trans[1]


class Box(CvxPolyhedron):
    """Six planes of a rectangular prism (geom/shape.py:135-210)."""

    def __init__(self, length: float, width: float, height: float, transform: Optional[RayTransform] = None,
                 l_grad: bool = False, w_grad: bool = False, h_grad: bool = False):
        super().__init__(transform=transform)
        hp = torch.pi / 2
        self.surfaces.append(_axis_plane([0.0, 0.0, length / 2], [0.0, 0.0, 0.0], l_grad))
        self.surfaces.append(_axis_plane([0.0, 0.0, -length / 2], [0.0, torch.pi, 0.0], l_grad))
        self.surfaces.append(_axis_plane([width / 2, 0.0, 0.0], [0.0, -hp, 0.0], w_grad))
        self.surfaces.append(_axis_plane([-width / 2, 0.0, 0.0], [0.0, hp, 0.0], w_grad))
        self.surfaces.append(_axis_plane([0.0, height / 2, 0.0], [hp, 0.0, 0.0], h_grad))
        self.surfaces.append(_axis_plane([0.0, -height / 2, 0.0], [-hp, 0.0, 0.0], h_grad))

    # extents between opposite faces (geom/shape.py:147-157)
    length = property(lambda self: self.surfaces[0].transform.trans[2] - self.surfaces[1].transform.trans[2])
    width = property(lambda self: self.surfaces[2].transform.trans[0] - self.surfaces[3].transform.trans[0])
    height = property(lambda self: self.surfaces[4].transform.trans[1] - self.surfaces[5].transform.trans[1])
