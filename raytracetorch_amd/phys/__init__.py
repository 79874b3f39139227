from .std import SurfaceFunction, Reflect, RefractSnell, RefractFresnel, Linear, Transmit, Block
from .filter import ApertureFilter, Fuzzy
# re-exported by the reference's star imports as well (phys/__init__.py)
from ..geom.transform import RayTransform  # noqa: E402,F401
