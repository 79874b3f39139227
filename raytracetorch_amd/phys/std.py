"""Surface physics functors (reference: phys/std.py:8-254).

A functor maps ``(hit_local, ray_dir, normal)`` to ``(new_dir, intensity_mult)``.  Inside a trace
that mapping is evaluated by the kernels from the ``phys_kind`` / ``phys`` fields each functor
writes into its surface record.  Calling a functor directly on ``[N,3]`` tensors runs the same
device code through ``rtt_apply_physics_f32``.
"""
import torch
import torch.nn as nn

from .. import kernels as K
from .. import program as P


class SurfaceFunction(nn.Module):

    _phys_kind = None

    def _fill_physics(self, rec):
        if self._phys_kind is None:
            raise NotImplementedError(f"{type(self).__name__} has no kernel description")
        rec["phys_kind"] = self._phys_kind

    def _diff_phys(self):
        return None

    def forward(self, local_intersect, ray_dir, normal, **kwargs):
        rec = P.new_record()
        self._fill_physics(rec)
        return K.apply_physics(rec, local_intersect, ray_dir, normal)


class Transmit(SurfaceFunction):
    """Direction unchanged, multiplier 1 (phys/std.py:221-235)."""
    _phys_kind = P.PHYS_TRANSMIT


class Block(SurfaceFunction):
    """Absorber: direction and multiplier zero (phys/std.py:237-254)."""
    _phys_kind = P.PHYS_BLOCK


class Reflect(SurfaceFunction):
    """d - 2 (d.n) n (phys/std.py:91-108)."""
    _phys_kind = P.PHYS_REFLECT


class RefractSnell(SurfaceFunction):
    """Snell refraction with total internal reflection (phys/std.py:111-145).

    `entering = d.n < 0`; the index ratio is `ior_out/ior_in` when entering and `ior_in/ior_out`
    otherwise; the output is not re-normalised."""
    _phys_kind = P.PHYS_SNELL

    def __init__(self, ior_in, ior_out, ior_in_grad=False, ior_out_grad=False):
        super().__init__()
        self.ior_in = nn.Parameter(torch.as_tensor(ior_in, dtype=torch.float32), requires_grad=ior_in_grad)
        self.ior_out = nn.Parameter(torch.as_tensor(ior_out, dtype=torch.float32), requires_grad=ior_out_grad)

    def _fill_physics(self, rec):
        rec["phys_kind"] = self._phys_kind
        n_in, n_out = P.scalar(self.ior_in), P.scalar(self.ior_out)
        rec["phys"][:] = (n_in, n_out, n_out / n_in, n_in / n_out)

    def _diff_phys(self):
        a, b = self.ior_in, self.ior_out
        return torch.stack([a, b, b / a, a / b])


class RefractFresnel(RefractSnell):
    """Stochastic refraction / reflection (phys/std.py:148-219): a ray is mirror-reflected with the
    unpolarised Fresnel reflectance R as probability (R = 1 under total internal reflection) and
    refracted by Snell's law otherwise; the multiplier is 1 either way.  The reference draws
    ``torch.rand_like``; the kernels draw Philox4x32-10 uniforms from a key this functor takes from
    torch's default generator each time a surface record is packed (reproducible under
    ``torch.manual_seed``, fresh for every trace)."""
    _phys_kind = P.PHYS_FRESNEL

    def _fill_physics(self, rec):
        super()._fill_physics(rec)
        rec["phys_kind"] = self._phys_kind
        key = int(torch.randint(0, 2 ** 62, (1,)).item())
        rec["seed"][:] = (key & 0xFFFFFFFF, key >> 32)


class Linear(SurfaceFunction):
    """Paraxial surface (phys/std.py:35-88): with the incoming direction expressed in the functor's frame
    and scaled to unit z, the outgoing slopes are (Cx x + Dx dx, Cy y + Dy dy); the result is normalised
    and rotated back.  ``transform`` supplies the frame (LinearElement installs the surface's own)."""
    _phys_kind = P.PHYS_LINEAR

    def __init__(self, Cx: float = 0, Cy: float = 0, Dx: float = 1, Dy=1, Cx_grad=False, Cy_grad=False,
                 Dx_grad=False, Dy_grad=False, transform=None):
        super().__init__()
        from ..geom.transform import RayTransform
        mk = lambda v, g: nn.Parameter(torch.as_tensor(v, dtype=torch.float32), requires_grad=g)
        self.Cx, self.Cy = mk(Cx, Cx_grad), mk(Cy, Cy_grad)
        self.Dx, self.Dy = mk(Dx, Dx_grad), mk(Dy, Dy_grad)
        self.transform = RayTransform() if transform is None else transform
        self._rot_is_surface = False   # set by LinearElement: rotate with the surface record's own Rj

    def _fill_physics(self, rec):
        rec["phys_kind"] = self._phys_kind
        rec["phys"][:] = (P.scalar(self.Cx), P.scalar(self.Cy), P.scalar(self.Dx), P.scalar(self.Dy))
        if self._rot_is_surface:
            rec["flags"] |= P.F_PHYS_ROT_SURF
        else:
            rec["phys_R"] = P.host_value(self.transform.rot).reshape(9)

    def _diff_phys(self):
        return torch.stack([self.Cx, self.Cy, self.Dx, self.Dy])
