from .transform import RayTransform, Vector3, Bool3, RayTransformBundle, RayTransformNoisy
from .primitives import Surface, Sphere, Quadric, Plane, QuadricZY, Cylinder, Cone
from .bounded import (SurfaceBounded, Rectangle, Disk, Ellipse, HalfSphere, BoundedHalfSphere, HalfCyl,
                      SingleCone)
from .shape import Shape, CvxPolyhedron, Box, Box4Side
from .spherics import Singlet, Doublet, Triplet
from .cylindrics import CylSinglet
