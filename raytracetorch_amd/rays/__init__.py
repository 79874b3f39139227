from .ray import Rays, Paths
from .bundle import Bundle, DiskSample, SolidAngleSample, CollimatedDisk, CollimatedLine, Fan, PointSource
from .panels import EmissionModel, LambertianEmission, SolidAngleEmission, PanelSource, RectangularPanel, RingSource
from .beam import GaussianBeam
from .particle import LambertianSphere, RayleighScatter, MieScatter
