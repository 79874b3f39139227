"""Flat area sources: a position sampler over the emitter and an emission model for the directions
(reference: rays/panels.py:10-154)."""
import math

import torch

from .bundle import AREA_RECT, AREA_RING, EMIT_CONE, EMIT_LAMBERT, SRC_PANEL, Bundle, DiskSample, SolidAngleSample


def lambert_local(u1, u2):
    """Cosine-weighted hemisphere about +z from two unit uniforms (rays/panels.py:27-37)."""
    r = torch.sqrt(u1)
    phi = 2.0 * math.pi * u2
    return torch.stack([r * torch.cos(phi), r * torch.sin(phi), torch.sqrt((1.0 - u1).clamp(min=0.0))], dim=1)


class EmissionModel:
    """Direction profile of a surface emitter (rays/panels.py:10-13)."""

    def sample_dir(self, N: int, device="cpu", dtype=torch.float32):
        raise NotImplementedError

    # (emission kind, parameters) of the device sampler, or None
    def _emission(self, dtype):
        return None


class LambertianEmission(EmissionModel):
    """p(theta) = cos(theta) / pi about +z (rays/panels.py:16-37)."""

    def sample_dir(self, N: int, device="cpu", dtype=torch.float32):
        u1 = torch.rand(N, device=device, dtype=dtype)
        u2 = torch.rand(N, device=device, dtype=dtype)
        return lambert_local(u1, u2)

    def _emission(self, dtype):
        return (EMIT_LAMBERT, ()) if type(self).sample_dir is LambertianEmission.sample_dir else None


class SolidAngleEmission(EmissionModel):
    """Uniform in solid angle inside a cone of half-angle `cone_angle` about +z (rays/panels.py:40-62)."""

    def __init__(self, cone_angle: float = math.pi / 4.0):
        self.cone_angle = cone_angle

    def _F_max(self, device, dtype):
        return SolidAngleSample.CDF_phi(torch.tensor(self.cone_angle, device=device, dtype=dtype))

    def sample_dir(self, N: int, device="cpu", dtype=torch.float32):
        mk = lambda v: torch.tensor(v, device=device, dtype=dtype)
        phi, theta = SolidAngleSample(mk(0.0), self._F_max(device, dtype), mk(0.0), mk(2.0 * math.pi)).sample(N)
        dr = torch.sin(phi)
        return torch.stack([torch.cos(theta) * dr, torch.sin(theta) * dr, torch.cos(phi)], dim=1)

    def _emission(self, dtype):
        if type(self).sample_dir is not SolidAngleEmission.sample_dir:
            return None
        return (EMIT_CONE, (0.0, float(self._F_max("cpu", dtype)), 0.0, 2.0 * math.pi))


class PanelSource(Bundle):
    """Area source: positions from the subclass, directions from `emission` about the local +z normal
    (rays/panels.py:65-83)."""

    def __init__(self, ray_id: int, device="cpu", dtype=torch.float32, transform=None, emission=None):
        super().__init__(ray_id=ray_id, device=device, dtype=dtype, transform=transform)
        self.emission = LambertianEmission() if emission is None else emission

    def sample_dir(self, N: int):
        return self.emission.sample_dir(N, device=self.device, dtype=self.dtype)

    def _source(self):
        return None   # no position sampler here

    def _panel_source(self, area_kind, par):
        emit = self.emission._emission(self.dtype) if isinstance(self.emission, EmissionModel) else None
        if emit is None:
            return None
        return (SRC_PANEL, par, area_kind, emit[0], emit[1])


class RectangularPanel(PanelSource):
    """Uniform over [-width/2, width/2] x [-height/2, height/2] in the local xy plane (rays/panels.py:86-122)."""

    def __init__(self, width: float, height: float, ray_id: int, device="cpu", dtype=torch.float32, transform=None,
                 emission=None):
        super().__init__(ray_id=ray_id, device=device, dtype=dtype, transform=transform, emission=emission)
        self.w2, self.h2 = float(width) / 2.0, float(height) / 2.0

    def sample_pos(self, N: int):
        x = self._uniform(-self.w2, self.w2, N)
        y = self._uniform(-self.h2, self.h2, N)
        return torch.stack([x, y, torch.zeros_like(x)], dim=1)

    def _source(self):
        return self._panel_source(AREA_RECT, (-self.w2, self.w2, -self.h2, self.h2))


class RingSource(PanelSource):
    """Uniform over the annulus radius_inner <= r <= radius_outer (rays/panels.py:125-154)."""

    def __init__(self, radius_inner: float, radius_outer: float, ray_id: int, device="cpu", dtype=torch.float32,
                 transform=None, emission=None):
        if radius_inner > radius_outer:
            raise ValueError(f"radius_inner ({radius_inner}) must be <= radius_outer ({radius_outer})")
        super().__init__(ray_id=ray_id, device=device, dtype=dtype, transform=transform, emission=emission)
        mk = lambda v: torch.tensor(v, device=device, dtype=dtype)
        self.r2 = (float(radius_inner) ** 2, float(radius_outer) ** 2)
        self._disk = DiskSample(mk(self.r2[0]), mk(self.r2[1]), mk(0.0), mk(2.0 * math.pi))

    def sample_pos(self, N: int):
        return self._disk.sample(N)

    def _source(self):
        return self._panel_source(AREA_RING, (self.r2[0], self.r2[1], 0.0, 2.0 * math.pi))
