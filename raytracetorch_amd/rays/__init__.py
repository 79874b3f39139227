from .ray import Rays, Paths
from .bundle import Bundle, DiskSample, SolidAngleSample, CollimatedDisk, CollimatedLine, Fan, PointSource
