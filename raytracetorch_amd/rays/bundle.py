"""Ray sources.  ``Bundle.sample(N)`` draws local positions / directions, places them with the
bundle transform and wraps them in ``Rays`` (reference: rays/bundle.py:9-171).

Sampling runs as ordinary torch ops on the bundle's device; it feeds the trace kernels and is not
part of them.
"""
import math
from typing import Optional, Union

import torch
import torch.nn as nn

from ..geom.transform import RayTransformBundle
from .ray import Rays


class Bundle(nn.Module):
    """Pencil of rays along +z from the origin of the bundle frame (rays/bundle.py:9-37)."""

    def __init__(self, ray_id: int, device: Union[str, torch.device] = "cpu",
                 dtype: torch.dtype = torch.float32, transform: Optional[RayTransformBundle] = None):
        super().__init__()
        self.ray_id = ray_id
        self.device = device
        self.dtype = dtype
        self.transform = RayTransformBundle(dtype=dtype) if transform is None else transform

    def _uniform(self, lo: float, hi: float, n: int):
        return torch.rand(n, device=self.device, dtype=self.dtype) * (hi - lo) + lo

    def sample_pos(self, N: int):
        return torch.zeros((N, 3), device=self.device, dtype=self.dtype)

    def sample_dir(self, N: int):
        d = torch.zeros((N, 3), device=self.device, dtype=self.dtype)
        d[:, 2] = 1.0
        return d

    def sample(self, N: int) -> Rays:
        p, d = self.transform.transform_(self.sample_pos(N), self.sample_dir(N))
        return Rays.initialize(p, d, ray_id=self.ray_id, device=self.device, dtype=self.dtype)


class DiskSample:
    """Uniform-in-area annular sector: r = sqrt(U(r_in^2, r_out^2)), theta = U(t0, t1)
    (rays/bundle.py:40-56)."""

    def __init__(self, radius_inner_2, radius_outer_2, theta_min, theta_max):
        self.r2 = (radius_inner_2, radius_outer_2)
        self.th = (theta_min, theta_max)

    def sample(self, N: int):
        lo, hi = self.th
        theta = torch.rand((N,), device=hi.device, dtype=hi.dtype) * (hi - lo) + lo
        lo, hi = self.r2
        r = torch.sqrt(torch.rand((N,), device=hi.device, dtype=hi.dtype) * (hi - lo) + lo)
        return torch.stack([r * torch.cos(theta), r * torch.sin(theta), torch.zeros_like(r)], dim=1)


class SolidAngleSample:
    """Polar angle with uniform solid-angle density, azimuth uniform (rays/bundle.py:58-80)."""

    def __init__(self, F_phi_min, F_phi_max, theta_min, theta_max):
        self.F = (F_phi_min, F_phi_max)
        self.th = (theta_min, theta_max)

    @classmethod
    def invCDF_phi(cls, F):
        return torch.acos(-2 * F + 1)

    @classmethod
    def CDF_phi(cls, phi):
        return (1 - torch.cos(phi)) / math.pi

    def sample(self, N: int):
        lo, hi = self.F
        phi = self.invCDF_phi(torch.rand((N,), device=hi.device, dtype=hi.dtype) * (hi - lo) + lo)
        lo, hi = self.th
        theta = torch.rand((N,), device=hi.device, dtype=hi.dtype) * (hi - lo) + lo
        return phi, theta


class CollimatedDisk(Bundle):
    """Collimated beam filling a disk of `radius` (rays/bundle.py:83-98)."""

    def __init__(self, radius: float, ray_id: int, device="cpu", dtype=torch.float32, transform=None):
        super().__init__(ray_id=ray_id, device=device, dtype=dtype, transform=transform)
        mk = lambda v: torch.tensor(v, device=device, dtype=dtype)
        self.radius2 = mk(radius * radius)
        self.disk = DiskSample(mk(0.0), self.radius2, mk(0.0), mk(2 * math.pi))

    def sample_pos(self, N: int):
        return self.disk.sample(N)


class CollimatedLine(Bundle):
    """Collimated beam along a segment of the local x axis (rays/bundle.py:101-118)."""

    def __init__(self, length: float, ray_id: int, device="cpu", dtype=torch.float32, transform=None):
        super().__init__(ray_id=ray_id, device=device, dtype=dtype, transform=transform)
        self.length_2 = float(length)

    def sample_pos(self, N: int):
        p = torch.zeros((N, 3), device=self.device, dtype=self.dtype)
        p[:, 0] = self._uniform(-self.length_2, self.length_2, N)
        return p


class Fan(Bundle):
    """Planar fan in the local yz plane with full opening `angle` (rays/bundle.py:121-140)."""

    def __init__(self, angle: float, ray_id: int, device="cpu", dtype=torch.float32, transform=None):
        super().__init__(ray_id=ray_id, device=device, dtype=dtype, transform=transform)
        self.angle_2 = float(angle) / 2

    def sample_dir(self, N: int):
        th = self._uniform(-self.angle_2, self.angle_2, N)
        return torch.stack([torch.zeros_like(th), torch.sin(th), torch.cos(th)], dim=1)


class PointSource(Bundle):
    """Diverging cone with numerical aperture `NA` (rays/bundle.py:143-171)."""

    def __init__(self, NA: float, ray_id: int, device="cpu", dtype=torch.float32, transform=None):
        super().__init__(ray_id=ray_id, device=device, dtype=dtype, transform=transform)
        mk = lambda v: torch.tensor(v, device=device, dtype=dtype)
        self.F_phi_max = SolidAngleSample.CDF_phi(torch.arcsin(mk(NA)))
        self.angle_dist = SolidAngleSample(mk(0.0), self.F_phi_max, mk(0.0), mk(2 * math.pi))

    def sample_dir(self, N: int):
        phi, theta = self.angle_dist.sample(N)
        dr = torch.sin(phi)
        return torch.stack([torch.cos(theta) * dr, torch.sin(theta) * dr, torch.cos(phi)], dim=1)
