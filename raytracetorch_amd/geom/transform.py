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


def _skew_exp(rot_vec: torch.Tensor) -> torch.Tensor:
    """geom/transform.py:52-57."""
    x, y, z = rot_vec
    K = torch.zeros((3, 3), device=rot_vec.device, dtype=rot_vec.dtype)
    K[0, 1], K[0, 2] = -z, y
    K[1, 0], K[1, 2] = z, -x
    K[2, 0], K[2, 1] = -y, x
    return torch.linalg.matrix_exp(K)


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
