"""Particle-like sources: a Lambertian sphere and scattering phase functions (reference: rays/particle.py)."""
import math

import torch

from .bundle import SRC_RAYLEIGH, SRC_SPHERE, Bundle
from .panels import lambert_local


class LambertianSphere(Bundle):
    """Uniform positions on a sphere of `radius`, cosine-weighted directions about the outward normal of each
    position (rays/particle.py:9-87).  Position and direction are coupled, so this class provides the joint
    local sample instead of the two separate hooks."""

    def __init__(self, radius: float, ray_id: int, device="cpu", dtype=torch.float32, transform=None):
        super().__init__(ray_id=ray_id, device=device, dtype=dtype, transform=transform)
        self.radius = float(radius)

    def _sample_local(self, N: int):
        rand = lambda: torch.rand(N, device=self.device, dtype=self.dtype)
        u1, u2 = rand(), rand()
        cos_t = 2.0 * u1 - 1.0
        sin_t = torch.sqrt((1.0 - cos_t ** 2).clamp(min=0.0))
        phi = 2.0 * math.pi * u2
        n = torch.stack([sin_t * torch.cos(phi), sin_t * torch.sin(phi), cos_t], dim=1)
        return n * self.radius, self._lambertian_around_normals(n, N)

    def _lambertian_around_normals(self, normals, N: int):
        """Rotate a +z hemisphere sample into the frame of each normal; the basis is Frisvad's (2012) branch-free
        construction (rays/particle.py:55-87)."""
        l = lambert_local(torch.rand(N, device=self.device, dtype=self.dtype),
                          torch.rand(N, device=self.device, dtype=self.dtype))
        nx, ny, nz = normals[:, 0], normals[:, 1], normals[:, 2]
        sign = torch.sign(nz + 1e-10)
        a = -1.0 / (sign + nz)
        b = nx * ny * a
        tangent = torch.stack([1.0 + sign * nx ** 2 * a, sign * b, -sign * nx], dim=1)
        bitangent = torch.stack([b, sign + ny ** 2 * a, -ny], dim=1)
        return l[:, 0:1] * tangent + l[:, 1:2] * bitangent + l[:, 2:3] * normals

    def _source(self):
        return (SRC_SPHERE, (self.radius,))


class RayleighScatter(Bundle):
    """Point scatterer with phase function p(theta) ~ 1 + cos^2(theta) about +z; cos(theta) by the closed-form
    inverse CDF, the real root of mu^3 + 3 mu - (8u - 4) = 0 (rays/particle.py:90-125)."""

    def sample_dir(self, N: int):
        u = torch.rand(N, device=self.device, dtype=self.dtype)
        half_q = 2.0 - 4.0 * u
        A = (torch.sqrt(half_q ** 2 + 1.0) - half_q).pow(1.0 / 3.0)
        cos_t = A - 1.0 / A
        sin_t = torch.sqrt((1.0 - cos_t ** 2).clamp(min=0.0))
        phi = 2.0 * math.pi * torch.rand(N, device=self.device, dtype=self.dtype)
        return torch.stack([sin_t * torch.cos(phi), sin_t * torch.sin(phi), cos_t], dim=1)

    def _source(self):
        return (SRC_RAYLEIGH, ())


class MieScatter(Bundle):
    """Holds the parameters of a Mie scatterer; like the reference it cannot sample directions
    (rays/particle.py:128-163)."""

    def __init__(self, particle_size_nm: float, wavelength_nm: float, particle_ior: float, environment_ior: float,
                 ray_id: int, device="cpu", dtype=torch.float32, transform=None):
        super().__init__(ray_id=ray_id, device=device, dtype=dtype, transform=transform)
        self.particle_size_nm = particle_size_nm
        self.wavelength_nm = wavelength_nm
        self.particle_ior = particle_ior
        self.environment_ior = environment_ior

    def sample_dir(self, N: int):
        raise NotImplementedError("MieScatter needs a Mie-theory solver for its phase function; override sample_dir()")
