"""Gaussian beam source (reference: rays/beam.py:9-41)."""
import torch

from .bundle import Bundle, SRC_GAUSS


class GaussianBeam(Bundle):
    """Collimated beam along +z whose x and y are independent normals with sigma = (1/e^2 diameter) / 4
    (rays/beam.py:28-41).  On a HIP device the normals come from the Box-Muller transform of the sampler's two
    Philox uniforms; on the CPU from torch.randn."""

    def __init__(self, diameter_1e2_x: float, diameter_1e2_y: float, ray_id: int, device="cpu",
                 dtype=torch.float32, transform=None):
        super().__init__(ray_id=ray_id, device=device, dtype=dtype, transform=transform)
        self.sigma_x = float(diameter_1e2_x) / 4.0
        self.sigma_y = float(diameter_1e2_y) / 4.0

    def sample_pos(self, N: int):
        x = torch.randn(N, device=self.device, dtype=self.dtype) * self.sigma_x
        y = torch.randn(N, device=self.device, dtype=self.dtype) * self.sigma_y
        return torch.stack([x, y, torch.zeros_like(x)], dim=1)

    def _source(self):
        return (SRC_GAUSS, (self.sigma_x, self.sigma_y))
