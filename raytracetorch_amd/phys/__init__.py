from .std import SurfaceFunction, Reflect, RefractSnell, Transmit, Block
from .filter import ApertureFilter
