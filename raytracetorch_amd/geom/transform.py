"""Rigid transforms (reference: geom/transform.py:10-139, :245-277).

``RayTransform`` only *holds* the pose (``trans``, ``rot_vec``) and produces the 3x3 rotation
``R = exp(skew(rot_vec))``; the per-ray products ``(p - t) @ R`` / ``d @ R`` of the trace happen
inside the HIP kernels, which receive R and t through the packed surface program.
``transform_`` / ``invTransform_`` remain as small tensor helpers for bundle placement.
"""
from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn

Vector3 = Optional[Union[torch.Tensor, List[float], Tuple[float, ...]]]
Bool3 = Optional[Union[torch.Tensor, List[bool], Tuple[bool, ...]]]


def _skew(rot_vec: torch.Tensor) -> torch.Tensor:
    x, y, z = rot_vec
    K = torch.zeros((3, 3), device=rot_vec.device, dtype=rot_vec.dtype)
    K[0, 1], K[0, 2] = -z, y
    K[1, 0], K[1, 2] = z, -x
    K[2, 0], K[2, 1] = -y, x
    return K


class _SkewExp(torch.autograd.Function):
    """R = exp(skew(v)) (geom/transform.py:52-57) with its derivative held ready.  The reference differentiates through
    `torch.linalg.matrix_exp`, whose backward takes the exponential of a 6x6 block matrix per upstream gradient; the
    slot Jacobians of a scene ask for hundreds of them at once (one per record slot, batched), which made a pose the
    most expensive thing to train.  Here the three directional derivatives dR/dv_k are computed once in the forward, as
    the upper right blocks of exp([[K, G_k], [0, K]]) (the Frechet derivative of the exponential along the generator
    G_k), and the backward is a contraction with them.  Same R bit for bit, same derivative."""
    generate_vmap_rule = True

    @staticmethod
    def forward(rot_vec):
        K = _skew(rot_vec)
        blocks = torch.zeros((3, 6, 6), device=rot_vec.device, dtype=rot_vec.dtype)
        blocks[:, :3, :3] = K
        blocks[:, 3:, 3:] = K
        blocks[0, 1, 5], blocks[0, 2, 4] = -1.0, 1.0     # G_x = d skew / d x
        blocks[1, 0, 5], blocks[1, 2, 3] = 1.0, -1.0     # G_y
        blocks[2, 0, 4], blocks[2, 1, 3] = -1.0, 1.0     # G_z
        return torch.linalg.matrix_exp(K), torch.linalg.matrix_exp(blocks)[:, :3, 3:].contiguous()

    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.save_for_backward(output[1])
        ctx.mark_non_differentiable(output[1])

    @staticmethod
    def backward(ctx, grad_R, _):
        (dR,) = ctx.saved_tensors
        return (grad_R.unsqueeze(0) * dR).sum(dim=(-2, -1))   # (plain ops: also under the batched backward of a Jacobian)


def _skew_exp(rot_vec: torch.Tensor) -> torch.Tensor:
    """geom/transform.py:52-57."""
    if rot_vec.requires_grad and torch.is_grad_enabled():
        return _SkewExp.apply(rot_vec)[0]
    return torch.linalg.matrix_exp(_skew(rot_vec))


class RayTransform(nn.Module):

    def __init__(self, rotation: Vector3 = None, translation: Vector3 = None,
                 dtype: torch.dtype = torch.float32,
                 trans_grad: bool = False, trans_mask: Bool3 = None,
                 rot_grad: bool = False, rot_mask: Bool3 = None):
        super().__init__()
        trans = torch.zeros(3, dtype=dtype) if translation is None else torch.as_tensor(translation)
        rot = torch.zeros(3) if rotation is None else torch.as_tensor(rotation)
        if not trans.is_floating_point():
            # the reference keeps an integer Parameter here (geom/transform.py:25) and relies on
            # type promotion in `pos - trans`; the promoted value is the float32 cast.
            trans = trans.to(dtype)
        if not rot.is_floating_point():
            rot = rot.to(dtype)
        self.trans = nn.Parameter(trans, requires_grad=trans_grad)
        self.rot_vec = nn.Parameter(rot, requires_grad=rot_grad)
        # gradient masks (geom/transform.py:29-46)
        if trans_grad and trans_mask is not None:
            self.register_buffer("trans_mask", torch.as_tensor(trans_mask, dtype=dtype))
            self.trans.register_hook(lambda g: g * self.trans_mask)
        if rot_grad and rot_mask is not None:
            self.register_buffer("rot_mask", torch.as_tensor(rot_mask, dtype=dtype))
            self.rot_vec.register_hook(lambda g: g * self.rot_mask)
        self._rot_cache = None

    @property
    def rot(self) -> torch.Tensor:
        """Local->global rotation; cached while `rot_vec` is frozen and unmodified
        (geom/transform.py:48-69)."""
        if self.rot_vec.requires_grad:
            return _skew_exp(self.rot_vec)
        tag = (self.rot_vec.data_ptr(), self.rot_vec._version)
        if self._rot_cache is None or self._rot_cache[0] != tag:
            self._rot_cache = (tag, _skew_exp(self.rot_vec.detach()))
        return self._rot_cache[1]

    # global -> local
    def transform(self, rays):
        return self.transform_(rays.pos, rays.dir)

    def transform_(self, _pos, _dir):
        R = self.rot
        return (_pos - self.trans[None, :]) @ R, _dir @ R

    # local -> global
    def invTransform(self, rays):
        return self.invTransform_(rays.pos, rays.dir)

    def invTransform_(self, _pos, _dir):
        Rt = self.rot.T
        return _pos @ Rt + self.trans[None, :], _dir @ Rt


def _affine5(column):
    """5x5 identity whose last column is `column` (geom/transform.py:121-139)."""
    eye = torch.eye(5, device=column.device, dtype=column.dtype)
    return torch.cat([eye[:, :4], column.unsqueeze(1)], dim=1)


def _paraxial(self):
    """Decentre and tilt of the element as an affine paraxial matrix acting on (x, u, y, v, 1):
    subtracts (trans_x, rot_x, trans_y, rot_y) (geom/transform.py:121-129)."""
    one = torch.ones((), device=self.trans.device, dtype=self.trans.dtype)
    return _affine5(-torch.stack([self.trans[0], self.rot_vec[0], self.trans[1], self.rot_vec[1], -one]))


def _paraxial_inv(self):
    """geom/transform.py:131-139."""
    one = torch.ones((), device=self.trans.device, dtype=self.trans.dtype)
    return _affine5(torch.stack([self.trans[0], self.rot_vec[0], self.trans[1], self.rot_vec[1], one]))


RayTransform.paraxial = _paraxial
RayTransform.paraxial_inv = _paraxial_inv


class RayTransformNoisy(RayTransform):
    """Pose with per-ray normal perturbations (geom/transform.py:141-242).  In the reference every `transform_` of
    this class raises IndexError (`self.rot_vec[:, 0]` on a 3-vector, geom/transform.py:184) and the constructor
    rejects a zero standard deviation, so there is no behaviour to reproduce: with all deviations zero this is a
    plain RayTransform, otherwise packing it into a surface program is refused."""

    def __init__(self, rotation: Vector3 = None, translation: Vector3 = None, dtype: torch.dtype = torch.float32,
                 std_translation: Vector3 = (0, 0, 0), std_rotation: Vector3 = (0, 0, 0),
                 trans_grad: bool = False, trans_mask: Bool3 = None, rot_grad: bool = False, rot_mask: Bool3 = None):
        super().__init__(rotation=rotation, translation=translation, dtype=dtype, trans_grad=trans_grad,
                         trans_mask=trans_mask, rot_grad=rot_grad, rot_mask=rot_mask)
        self.trans_scale = nn.Parameter(torch.as_tensor(std_translation, dtype=dtype), requires_grad=False)
        self.rot_scale = nn.Parameter(torch.as_tensor(std_rotation, dtype=dtype), requires_grad=False)
        self.is_frozen = False

    @property
    def noisy(self) -> bool:
        return bool((self.trans_scale != 0).any() or (self.rot_scale != 0).any())

    def addNoise(self, N):
        """Per-ray (translation [N,3], rotation vector [N,3]); without deviations every row is the pose itself
        (geom/transform.py:195-200)."""
        return self.trans[None, :].expand(N, 3), self.rot_vec[None, :].expand(N, 3)

    def freeze_noise(self, N):
        self.is_frozen = True

    def unfreeze_noise(self):
        self.is_frozen = False


class RayTransformBundle(RayTransform):
    """Pose of a ray source: `transform_` maps the bundle frame to the global frame
    (geom/transform.py:245-277)."""

    def transform_(self, _pos, _dir):
        Rt = self.rot.T
        return _pos @ Rt + self.trans[None, :], _dir @ Rt

    def invTransform_(self, _pos, _dir):
        R = self.rot
        return (_pos - self.trans[None, :]) @ R, _dir @ R
