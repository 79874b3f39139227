"""Aperture stop physics (reference: phys/filter.py:10-33).

``ApertureFilter(inBounds)`` transmits rays whose local hit point lies inside the bound and zeroes
direction and intensity multiplier of the others.  The bound must be the ``inBounds`` method of a
bounded surface of this package so that it can be described to the kernels; it is evaluated
*without* that surface's ``invert`` flag, exactly as the bound method is in the reference.
"""
from typing import Callable

from .. import program as P
from .std import Transmit


class ApertureFilter(Transmit):
    _phys_kind = P.PHYS_APERTURE

    def __init__(self, inBounds: Callable):
        super().__init__()
        self._inBounds = inBounds

    def _bound_owner(self):
        owner = getattr(self._inBounds, "__self__", None)
        if owner is None or not hasattr(owner, "_fill_bounds"):
            raise NotImplementedError(
                "ApertureFilter needs the inBounds method of a bounded surface (Disk, Rectangle, "
                "Ellipse, ...); arbitrary callables cannot run inside the HIP kernels")
        return owner

    def _fill_physics(self, rec):
        rec["phys_kind"] = self._phys_kind
        owner = self._bound_owner()
        probe = P.new_record()
        owner._fill_bounds(probe)
        if int(rec["bound_kind"]) == P.BND_NONE:
            # functor used on its own: take bound (and curvature for cap bounds) from the owner
            owner._fill_geometry(rec)
            owner._fill_bounds(rec)
        elif int(probe["bound_kind"]) != int(rec["bound_kind"]) or not (probe["bnd"] == rec["bnd"]).all():
            raise NotImplementedError("ApertureFilter bound differs from the bound of the surface it sits on")


class Fuzzy(Transmit):
    """Transmit with an intensity multiplier given by a user function of the local hit point
    (phys/filter.py:36-50).  A Python callable cannot run inside a kernel; geometry does not depend on it, so the
    kernels treat the surface as a Transmit whose local hit points are recorded (like a sensor's), and the scene
    applies `intensity_function(hit_local)` to the rays that hit it afterwards: to the final intensities and to the
    intensity every later sensor recorded (sequential scenes), or after each bounce (nearest-hit scenes, which
    then run one launch per bounce).  No gradients flow through the function."""
    _phys_kind = P.PHYS_TRANSMIT

    def __init__(self, intensity_function: Callable):
        super().__init__()
        self.block_function = intensity_function

    def _fill_physics(self, rec):
        rec["phys_kind"] = self._phys_kind
        rec["flags"] |= P.F_SENSOR   # have the kernels record hit_local for the host-side function

    def forward(self, local_intersect, ray_dir, normal, **kwargs):
        out_dir, _ = super().forward(local_intersect, ray_dir, normal)
        return out_dir, self.block_function(local_intersect)
