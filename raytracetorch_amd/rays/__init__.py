from .ray import Rays
from .bundle import Bundle, DiskSample, SolidAngleSample, CollimatedDisk, CollimatedLine, Fan, PointSource
