"""``Element`` = a shape plus one physics functor per surface (reference: elements/parent.py:8-81).

``intersectTest(rays) -> [N, K]`` and ``forward(rays, j) -> (new_pos, new_dir, intensity_mult)``
keep the reference contract; both run in the HIP kernels from the element's surface records.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import kernels as K
from .. import program as P
from ..geom import Shape
from ..phys import SurfaceFunction


class Element(nn.Module):

    def __init__(self):
        super().__init__()
        self.shape = Shape()
        self.surface_functions = nn.ModuleList()

    # ---- record description -------------------------------------------------------------------
    def _record(self, j: int):
        rec = self.shape._record(j)
        self.surface_functions[j]._fill_physics(rec)
        return rec

    def _program(self):
        return np.stack([self._record(j) for j in range(len(self.shape))])

    def _diff_slots(self, j: int):
        d = self.shape._diff_slots(j)
        d["phys"] = self.surface_functions[j]._diff_phys()
        return d

    # ---- reference API ------------------------------------------------------------------------
    def intersectTest(self, rays):
        return K.intersect_test(rays, self._program())

    def _forward_full(self, rays, surf_idx):
        return K.surface_forward(rays, self._record(surf_idx)[None], 0, want_phys=True)

    def forward(self, rays, surf_idx):
        """(hit_global [N,3], new_dir [N,3], intensity multiplier [N]) of surface `surf_idx` for every ray
        (elements/parent.py:44-58).  Under grad mode the first two are differentiable with respect to the element's
        trainable parameters and to the rays (tests/test_optimize_singlet.py:66-116 of the reference chains two such
        calls and back-propagates a spot loss)."""
        o = self._forward_full(rays, surf_idx)
        fn = self.surface_functions[surf_idx]
        if hasattr(fn, "block_function"):   # Fuzzy: the multiplier is a host-side function of the local hit point
            return o["hit_global"], o["new_dir"], o["imult"] * fn.block_function(o["hit_local"])
        from .. import autograd as AG
        if AG.needs_grad([self]) or AG.rays_carry_grad(rays):
            # the same visit as a one-surface sequential trace, whose backward exists; like Shape.forward it applies no
            # shape-level bound (geom/shape.py:61-87), and rays without a hit keep what the plain call returned
            rec = self._record(surf_idx)
            rec["shape_kind"] = P.SHP_NONE
            rec["sensor_slot"] = -1
            rec["flags"] &= ~P.F_SENSOR
            outs = AG.trace_sequential_with_grad([_OneSurface(self, surf_idx)], rays, rec[None])
            hit = (o["t"] < float("inf"))[:, None]
            return (torch.where(hit, outs[0].t(), o["hit_global"]), torch.where(hit, outs[1].t(), o["new_dir"]),
                    o["imult"])
        return o["hit_global"], o["new_dir"], o["imult"]

    # ---- paraxial description (host-side 5x5 matrices on (x, u, y, v, 1); no rays involved) -----------
    def _paraxial(self):
        t = self.shape.transform.trans
        return torch.eye(5, device=t.device, dtype=t.dtype)

    def _posed(self, mat):
        """Matrix of a surface seen from the global axis: undo the element's decentre / tilt, act, redo."""
        tf = self.shape.transform
        return tf.paraxial_inv() @ mat @ tf.paraxial()

    def getParaxial(self):
        """([z of each paraxial surface], [its 5x5 matrix]) (elements/parent.py:60-70)."""
        return [self.shape.z], [self._posed(self._paraxial())]


class _OneSurface:
    """One surface of an element where the gradient code expects an element with a one-surface shape."""

    class _Len1:
        def __len__(self):
            return 1

    def __init__(self, element, j):
        self.element, self.j, self.shape = element, j, self._Len1()

    def parameters(self):
        return self.element.parameters()

    def modules(self):
        return self.element.modules()

    def _diff_slots(self, _j):
        return self.element._diff_slots(self.j)


class ElementCustom(Element):
    """One functor shared by every surface of a given shape (elements/parent.py:73-81)."""

    def __init__(self, shape: Shape, surface_function: SurfaceFunction, device: torch.device = None):
        super().__init__()
        self.shape = shape
        self.surface_functions.extend(len(shape) * [surface_function])
