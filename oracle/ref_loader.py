"""Container-only loader for the upstream reference (TEST INFRASTRUCTURE, never shipped or imported
by the product package).

The reference at ``/root/reference`` cannot be imported as a package: ``rays/ray.py:3`` needs
``tensordict`` (absent, no network) and ``rays/__init__.py:11`` imports a name that
``rays/panels.py`` does not define.  This module leaves every reference file untouched and

1. registers a minimal stand-in for ``tensordict.tensorclass`` that covers exactly what
   ``rays/ray.py:7-97`` uses (keyword ``__init__`` with ``batch_size``, ``__post_init__``, attribute
   get/set, boolean-mask ``__getitem__`` and ``.to`` that do *not* re-run ``__post_init__``);
2. registers stub package objects ``RayTraceTorch`` and ``RayTraceTorch.rays`` whose ``__path__``
   point into the reference tree, so the broken package ``__init__`` files are never executed.

It is used only by ``oracle/make_goldens.py`` (fixture generation) and by the optional
``tests/test_oracle_vs_reference.py`` cross-check, both of which skip when the reference tree is
not present (it never is on the GPU box).
"""
import importlib
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("RTT_REFERENCE_ROOT", "/root/reference")


def _tensorclass(cls):
    fields = list(cls.__annotations__)
    post = getattr(cls, "__post_init__", None)

    def __init__(self, *, batch_size, **kw):
        for f in fields:
            object.__setattr__(self, f, kw[f])
        object.__setattr__(self, "batch_size", torch.Size(batch_size))
        if post is not None:
            post(self)

    def _from(vals, bs):
        o = cls.__new__(cls)
        for f, v in zip(fields, vals):
            object.__setattr__(o, f, v)
        object.__setattr__(o, "batch_size", torch.Size(bs))
        return o

    def __getitem__(self, idx):
        vals = [getattr(self, f)[idx] for f in fields]
        return _from(vals, [vals[0].shape[0]])

    def to(self, *a, **k):
        return _from([getattr(self, f).to(*a, **k) for f in fields], self.batch_size)

    cls.__init__ = __init__
    cls.__getitem__ = __getitem__
    cls.to = to
    return cls


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "geom"))


def load():
    """Import the reference sub-packages and return the stub top-level module."""
    if "RayTraceTorch" in sys.modules and getattr(sys.modules["RayTraceTorch"], "_rtt_stub", False):
        return sys.modules["RayTraceTorch"]
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")

    td = types.ModuleType("tensordict")
    td.tensorclass = _tensorclass
    sys.modules.setdefault("tensordict", td)

    top = types.ModuleType("RayTraceTorch")
    top.__path__ = [REFERENCE_ROOT]
    top._rtt_stub = True
    sys.modules["RayTraceTorch"] = top

    rays = types.ModuleType("RayTraceTorch.rays")
    rays.__path__ = [os.path.join(REFERENCE_ROOT, "rays")]
    sys.modules["RayTraceTorch.rays"] = rays
    top.rays = rays

    top.geom = importlib.import_module("RayTraceTorch.geom")
    ray_mod = importlib.import_module("RayTraceTorch.rays.ray")
    rays.Rays, rays.Paths = ray_mod.Rays, ray_mod.Paths
    bun = importlib.import_module("RayTraceTorch.rays.bundle")
    for n in ("Bundle", "DiskSample", "SolidAngleSample", "CollimatedDisk", "CollimatedLine", "Fan",
              "PointSource"):
        setattr(rays, n, getattr(bun, n))
    for mod, names in (("panels", ("EmissionModel", "LambertianEmission", "SolidAngleEmission", "PanelSource",
                                   "RectangularPanel", "RingSource")),
                       ("beam", ("GaussianBeam",)),
                       ("particle", ("LambertianSphere", "RayleighScatter", "MieScatter"))):
        m = importlib.import_module("RayTraceTorch.rays." + mod)
        for n in names:
            setattr(rays, n, getattr(m, n))
    top.phys = importlib.import_module("RayTraceTorch.phys")
    top.elements = importlib.import_module("RayTraceTorch.elements")
    top.scene = importlib.import_module("RayTraceTorch.scene")
    top.optim = importlib.import_module("RayTraceTorch.optim")
    top.render = importlib.import_module("RayTraceTorch.render.camera")
    top.project = importlib.import_module("RayTraceTorch.project")
    return top
