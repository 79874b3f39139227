from .parent import Element, ElementCustom
from .lens import SingletLens, CylSingletLens, DoubletLens, TripletLens
from .sensor import Sensor
from .mirror import Mirror, SphericalMirror, CylindricalMirror, ParabolicMirror, ParabolicMirrorXZ, ParabolicMirrorOffAxis
from .aperture import CircularAperture, RectangularAperture, EllipticAperture
from .ideal import (LinearElement, IdealThinLens, IdealCylThinLens, IdealMirror, ParaxialLensMat, ParaxialDistMat,
                    ParaxialRefractMat, ParaxialMirrorMat)
