"""Cylindrical lens stacks: two ``HalfCyl`` faces curved along y plus four side planes
(reference: geom/cylindrics.py:10-119).

Shape-level validity (geom/cylindrics.py:23-55), element frame, surfaces ordered
[front, back, +x, -x, +y, -y]:
  all    : x_min-1e-5 <= x <= x_max+1e-5  and  y_min-1e-5 <= y <= y_max+1e-5
  sides  : additionally  sag_front(y)+1e-4 <= z <= sag_back(y)-1e-4  with the ray's own y
"""
from typing import Optional

import torch

from .. import program as P
from .bounded import HalfCyl
from .shape import Box4Side, Shape
from .transform import RayTransform


class Cylindric(Shape):

    def _make_cyl_surface(self, C, z_vertex, c_grad=False, z_grad=False):
        t = RayTransform(translation=[0., 0., z_vertex], trans_grad=z_grad, trans_mask=[False, False, True])
        return HalfCyl(curvature=C, transform=t, curvature_grad=c_grad)

    def _limits(self):
        s = self.surfaces
        return (s[2].transform.trans[0], s[3].transform.trans[0],
                s[4].transform.trans[1], s[5].transform.trans[1])

    def _fill_shape_bound(self, rec, j: int):
        s = self.surfaces
        m = P.f32(1e-5)
        x_max = P.host_value(s[2].transform.trans)[0]
        x_min = P.host_value(s[3].transform.trans)[0]
        y_max = P.host_value(s[4].transform.trans)[1]
        y_min = P.host_value(s[5].transform.trans)[1]
        rec["shp"][:4] = (x_max + m, x_min - m, y_max + m, y_min - m)
        if j < self.N_optical:
            rec["shape_kind"] = P.SHP_CYL_FACE
        else:
            rec["shape_kind"] = P.SHP_CYL_EDGE
            rec["shp"][4:] = (P.scalar(s[0].c), P.host_value(s[0].transform.trans)[2],
                              P.scalar(s[1].c), P.host_value(s[1].transform.trans)[2])

    def inBounds(self, local_pos, surf_idx):
        x, y, z = local_pos[:, 0], local_pos[:, 1], local_pos[:, 2]
        x_max, x_min, y_max, y_min = self._limits()
        ap = (x <= x_max + 1e-5) & (x >= x_min - 1e-5) & (y <= y_max + 1e-5) & (y >= y_min - 1e-5)
        if surf_idx < self.N_optical:
            return ap
        zf, zb = self.surfaces[0].sagittalZ(y), self.surfaces[1].sagittalZ(y)
        return (z >= zf + 1e-4) & (z <= zb - 1e-4) & ap


class CylSinglet(Cylindric):
    """geom/cylindrics.py:58-119."""

    def __init__(self, C1: float, C2: float, width: float, height: float, T: float,
                 C1_grad: bool = True, C2_grad: bool = True, T_grad: bool = True,
                 w_grad: bool = False, h_grad: bool = False, transform: Optional[RayTransform] = None):
        super().__init__(transform=transform)
        self.N_optical = 2
        self.surfaces.append(self._make_cyl_surface(C1, -T / 2, c_grad=C1_grad, z_grad=T_grad))
        self.surfaces.append(self._make_cyl_surface(C2, T / 2, c_grad=C2_grad, z_grad=T_grad))
        for s in Box4Side(width, height, w_grad=w_grad, h_grad=h_grad).surfaces:
            self.surfaces.append(s)
        with torch.no_grad():
            if abs(0.5 * C1) > 1 / height:
                raise ValueError("|R1| must be larger than Height/2")
            if abs(0.5 * C2) > 1 / height:
                raise ValueError("|R2| must be larger than Height/2")
            if self.surfaces[0].sagittalZ(torch.as_tensor(height / 2)) > self.surfaces[1].sagittalZ(torch.as_tensor(height / 2)):
                raise ValueError("Front and back surfaces intersecting")

    @property
    def width(self):
        return self.surfaces[2].transform.trans[0] - self.surfaces[3].transform.trans[0]

    @property
    def height(self):
        return self.surfaces[4].transform.trans[1] - self.surfaces[5].transform.trans[1]
