from .parent import Element, ElementCustom
from .lens import SingletLens, CylSingletLens, DoubletLens, TripletLens
from .sensor import Sensor
from .mirror import Mirror, SphericalMirror, CylindricalMirror, ParabolicMirror, ParabolicMirrorXZ, ParabolicMirrorOffAxis
from .aperture import CircularAperture, RectangularAperture, EllipticAperture
from .ideal import (LinearElement, IdealThinLens, IdealCylThinLens, IdealMirror, ParaxialLensMat, ParaxialDistMat,
                    ParaxialRefractMat, ParaxialMirrorMat)
# The reference's `from .lens import *` / `.mirror` / `.aperture` (elements/__init__.py:3-6) also re-export what those
# modules import, and its scripts rely on it (`rtt.elements.CylSinglet`, tests/test_cylindrics.py:106): same names here.
from ..geom import (RayTransform, Singlet, CylSinglet, Doublet, Triplet, Quadric, QuadricZY, Disk, Rectangle,  # noqa: E402,F401
                    Ellipse)
from ..geom.bounded import HalfSphere, HalfCyl, BoundedHalfSphere  # noqa: E402,F401
from ..phys import RefractSnell, RefractFresnel, Block, Reflect, ApertureFilter  # noqa: E402,F401
