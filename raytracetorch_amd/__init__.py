"""raytracetorch_amd — MI355X (gfx950) implementation of RayTraceTorch's batched ray-propagation
path behind the reference's Python API (rays / geom / phys / elements / scene / optim)."""
from . import rays, geom, phys, elements, scene, optim, render
from .rays import Rays
from .scene import Scene, SequentialScene
from .project import load_scene, load_project, save_project

__all__ = ["rays", "geom", "phys", "elements", "scene", "optim", "render", "Rays", "Scene", "SequentialScene", "load_scene", "load_project",
           "save_project"]
