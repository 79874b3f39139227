"""``Rays``: the ray-bundle container, stored structure-of-arrays.

Call-compatible with the reference's tensorclass (rays/ray.py:7-97): fields ``pos [N,3]``,
``dir [N,3]``, ``intensity [N]``, ``id [N] int8``, ``wavelength [N]``, ``batch_size``; methods
``initialize``, ``scatter_update``, ``with_coords``, mask ``__getitem__`` and ``to``.

Storage differs on purpose: positions and directions live as contiguous ``[3, N]`` planes so that a
wavefront's 64 lanes read 256 contiguous bytes per component.  ``.pos`` / ``.dir`` are ``[N, 3]``
transposed *views* of those planes, so reference-style indexing keeps working without copies.
"""
from typing import Union

import torch


def _to_planes(v: torch.Tensor) -> torch.Tensor:
    """[N,3] -> contiguous [3,N]; free when `v` already is a transposed view of such planes."""
    if v.ndim != 2 or v.shape[1] != 3:
        raise ValueError(f"expected an [N, 3] tensor, got {tuple(v.shape)}")
    return v.t().contiguous()


def _normalize_planes(d: torch.Tensor) -> torch.Tensor:
    """rays/ray.py:22-25: F.normalize(dir, p=2, dim=1) with eps 1e-12, on [3,N] planes."""
    n = torch.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
    return d / torch.clamp_min(n, 1e-12)


class Rays:
    __slots__ = ("_pos", "_dir", "intensity", "id", "wavelength")

    def __init__(self, pos, dir, intensity, id, wavelength, batch_size=None):
        self._pos = _to_planes(pos)
        self._dir = _normalize_planes(_to_planes(dir))
        self.intensity = intensity
        self.id = id
        self.wavelength = wavelength
        if batch_size is not None and int(list(batch_size)[0]) != self._pos.shape[1]:
            raise ValueError("batch_size does not match the number of rays")

    # ---- construction without the transposition copy / normalisation ------------------------
    @classmethod
    def from_planes(cls, pos_planes, dir_planes, intensity, id, wavelength, normalize=False):
        """Wrap existing contiguous [3,N] planes (no copy).  `normalize=True` applies the
        constructor's direction normalisation."""
        r = object.__new__(cls)
        if not (pos_planes.is_contiguous() and dir_planes.is_contiguous()):
            raise ValueError("planes must be contiguous [3, N] tensors")
        r._pos = pos_planes
        r._dir = _normalize_planes(dir_planes) if normalize else dir_planes
        r.intensity, r.id, r.wavelength = intensity, id, wavelength
        return r

    @classmethod
    def initialize(cls, origins, directions, wavelengths=None, intensities=None, ray_id: int = 0,
                   device: Union[str, torch.device] = "cpu", dtype: torch.dtype = torch.float32):
        """rays/ray.py:42-82."""
        origins = torch.as_tensor(origins, device=device, dtype=dtype)
        directions = torch.as_tensor(directions, device=device, dtype=dtype)
        if origins.ndim == 1:
            origins = origins.unsqueeze(0)
        if directions.ndim == 1:
            directions = directions.unsqueeze(0)
        n = origins.shape[0]
        if intensities is None:
            intensities = torch.ones(n, device=device, dtype=dtype)
        else:
            intensities = torch.as_tensor(intensities, device=device, dtype=dtype)
        if wavelengths is None:
            wavelengths = torch.zeros(n, device=device, dtype=dtype)
        else:
            wavelengths = torch.as_tensor(wavelengths, device=device, dtype=dtype)
        ids = torch.full((n,), ray_id, dtype=torch.int8, device=device)
        return cls(pos=origins, dir=directions, intensity=intensities, id=ids, wavelength=wavelengths,
                   batch_size=[n])

    # ---- [N,3] views -------------------------------------------------------------------------
    @property
    def pos(self):
        return self._pos.t()

    @pos.setter
    def pos(self, v):
        self._pos = _to_planes(v)

    @property
    def dir(self):
        return self._dir.t()

    @dir.setter
    def dir(self, v):
        self._dir = _to_planes(v)

    @property
    def batch_size(self):
        return torch.Size([self._pos.shape[1]])

    @property
    def device(self):
        return self._pos.device

    def __len__(self):
        return self._pos.shape[1]

    # ---- reference methods -------------------------------------------------------------------
    def scatter_update(self, mask, new_pos, new_dir, intensity_mod):
        """rays/ray.py:29-40 (out-of-place, like index_put)."""
        pos, dir_, inten = self._pos.clone(), self._dir.clone(), self.intensity.clone()
        pos[:, mask] = new_pos.t()
        dir_[:, mask] = new_dir.t()
        inten[mask] = inten[mask] * intensity_mod
        self._pos, self._dir, self.intensity = pos, dir_, inten

    def with_coords(self, new_pos, new_dir):
        """rays/ray.py:84-97 (runs the constructor normalisation again)."""
        return Rays(pos=new_pos, dir=new_dir, intensity=self.intensity, id=self.id,
                    wavelength=self.wavelength, batch_size=self.batch_size)

    def __getitem__(self, idx):
        return Rays.from_planes(self._pos[:, idx].contiguous(), self._dir[:, idx].contiguous(),
                                self.intensity[idx], self.id[idx], self.wavelength[idx])

    def to(self, *args, **kwargs):
        return Rays.from_planes(self._pos.to(*args, **kwargs), self._dir.to(*args, **kwargs),
                                self.intensity.to(*args, **kwargs), self.id.to(*args, **kwargs),
                                self.wavelength.to(*args, **kwargs))

    def clone(self):
        return Rays.from_planes(self._pos.clone(), self._dir.clone(), self.intensity.clone(),
                                self.id.clone(), self.wavelength.clone())

    def __repr__(self):
        return f"Rays(N={len(self)}, device={self.device}, layout=planes[3,N])"


class PendingSample:
    """What a device sampler call would generate: the source description of the C interface, the number of rays and the
    device."""
    __slots__ = ("src", "n", "device")

    def __init__(self, src, n, device):
        self.src, self.n, self.device = src, n, device

    def make(self):
        from .. import kernels as K
        return K.generate_sample(self)


class PendingTrace:
    """The result of a sequential trace that has not been written: the input bundle, the packed program of the call
    (with the keys of its random draws, so the same branches are taken) and the fields a trace leaves alone."""
    __slots__ = ("src", "prog", "ids", "wavelength", "n", "device")

    def __init__(self, src, prog, ids, wavelength):
        self.src, self.prog, self.ids, self.wavelength = src, prog, ids, wavelength
        self.n, self.device = len(src), src.device

    def make(self):
        from .. import kernels as K
        out = Rays.from_planes(torch.empty_like(self.src._pos), torch.empty_like(self.src._dir),
                               torch.empty_like(self.src.intensity), self.ids, self.wavelength)
        K.trace_sequential_record(self.src, self.prog, out=out)   # the record of this repeat is not delivered to anybody
        return out._pos, out._dir, out.intensity, self.ids, self.wavelength


_SLOTS = {name: Rays.__dict__[name] for name in Rays.__slots__}


class SampledRays(Rays):
    """Rays whose planes are still to be made.  The rays of a `Bundle.sample(N)` on a HIP device, before anybody has
    looked at them: the call has fixed which rays these are (source, pose, Philox key and global offset), but the planes
    are generated only when they are first read — or never: `SequentialScene.simulate` generates and traces such rays in
    one launch, which writes the traced planes and reads nothing.  Likewise the traced bundle behind a fused spot-size
    loss, which only reads the sensor (`PendingTrace`).  The first access turns the object into plain `Rays`; until then
    the description sits in the `wavelength` slot, so both classes share one layout."""
    __slots__ = ()

    @classmethod
    def pending(cls, src, n, device):
        r = object.__new__(cls)
        _SLOTS["wavelength"].__set__(r, PendingSample(src, n, device))
        return r

    @classmethod
    def defer(cls, rays, pending):
        """Turn the plain `rays` object into one whose planes `pending.make()` provides when they are first read."""
        for name in ("_pos", "_dir", "intensity", "id"):
            _SLOTS[name].__set__(rays, None)
        _SLOTS["wavelength"].__set__(rays, pending)
        rays.__class__ = cls
        return rays

    def _pending(self):
        return _SLOTS["wavelength"].__get__(self)

    def _take(self):
        """Become plain `Rays` whose slots the caller fills, and hand out the description."""
        pend = _SLOTS["wavelength"].__get__(self)
        self.__class__ = Rays
        return pend

    def _force(self):
        planes = self._pending().make()   # should this raise, the object stays the pending bundle it was
        self._take()
        self._pos, self._dir, self.intensity, self.id, self.wavelength = planes

    @property
    def batch_size(self):
        return torch.Size([_SLOTS["wavelength"].__get__(self).n])

    @property
    def device(self):
        return _SLOTS["wavelength"].__get__(self).device

    def __len__(self):
        return _SLOTS["wavelength"].__get__(self).n


def _lazy_slot(name):
    slot = _SLOTS[name]

    def get(self):
        self._force()
        return slot.__get__(self)

    def put(self, v):
        self._force()
        slot.__set__(self, v)
    return property(get, put)


for _name in Rays.__slots__:
    setattr(SampledRays, _name, _lazy_slot(_name))


class Paths:
    """Proxy around ``Rays`` that keeps the position of every ray after each update, for visualisation
    (rays/ray.py:100-207).  ``Scene.step`` and ``SequentialScene.simulate`` recognise the wrapper: a bounce of
    the nearest-hit tracer adds one snapshot, a sequential trace adds one per surface that was hit
    (``rtt_trace_seq_history_f32``), as the reference's per-surface ``scatter_update`` calls do."""

    def __init__(self, rays: Rays):
        self._rays = rays
        self._history = [rays.pos.clone().detach().cpu()]

    pos = property(lambda self: self._rays.pos, lambda self, v: setattr(self._rays, "pos", v))
    dir = property(lambda self: self._rays.dir, lambda self, v: setattr(self._rays, "dir", v))
    intensity = property(lambda self: self._rays.intensity, lambda self, v: setattr(self._rays, "intensity", v))
    id = property(lambda self: self._rays.id)
    wavelength = property(lambda self: self._rays.wavelength)
    batch_size = property(lambda self: self._rays.batch_size)
    device = property(lambda self: self._rays.device)

    def __len__(self):
        return len(self._rays)

    def __getitem__(self, idx):
        return self._rays[idx]

    def _snapshot(self, planes=None):
        src = self._rays._pos if planes is None else planes
        self._history.append(src.t().clone().detach().cpu())

    def scatter_update(self, mask, new_pos, new_dir, intensity_mod):
        self._rays.scatter_update(mask, new_pos, new_dir, intensity_mod)
        self._snapshot()

    def unwrap(self) -> Rays:
        return self._rays

    def get_history(self) -> list:
        """[N, 3] CPU tensors: the initial positions, then one snapshot per recorded update."""
        return self._history

    def to(self, device):
        self._rays = self._rays.to(device)
        return self
