from .std import SurfaceFunction, Reflect, RefractSnell, RefractFresnel, Linear, Transmit, Block
from .filter import ApertureFilter, Fuzzy
