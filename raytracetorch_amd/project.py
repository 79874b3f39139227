"""``.rtt`` project files: JSON of element / bundle configurations (reference: project.py:45-179).

    save_project(path, element_configs, bundle_configs, settings)     configurations -> file
    load_project(path) -> {'version', 'settings', 'elements', 'bundles'}
    load_scene(path, device=..., dtype=..., sample_rays=True, nbounces=None) -> Scene on the HIP path

``load_scene`` returns a ``Scene`` whose ``step`` / ``simulate`` run in the kernels; ``scene.to_sequential()``
gives the fixed-order variant.  The device stored in a file is ignored, as in the reference.
"""
import json
import os
from typing import Optional, Union

import torch
import torch.nn as nn

from .config import instantiate_from_config
from .scene import Scene

PROJECT_VERSION = "1.0"


def save_project(path: str, element_configs: list, bundle_configs: list, settings: Optional[dict] = None) -> None:
    """project.py:45-63"""
    with open(path, "w", encoding="utf-8") as f:
        json.dump({"version": PROJECT_VERSION, "settings": settings or {}, "elements": element_configs,
                   "bundles": bundle_configs}, f, indent=2)


def migrate_project(data: dict) -> dict:
    """project.py:79-89: only schema 1.0 exists."""
    version = data.get("version", "0.0")
    if version != "1.0":
        raise ValueError(f"Unsupported project version: {version}")
    return data


def load_project(path: str) -> dict:
    """project.py:66-76"""
    with open(path, "r", encoding="utf-8") as f:
        data = json.load(f)
    if "version" not in data:
        raise ValueError(f"Not a valid .rtt project file: {os.path.basename(path)}")
    return migrate_project(data)


def _lookup(modules, names, target, kind):
    for module, name in zip(modules, names):
        if name == target:
            return module
    raise KeyError(f"No {kind} named {target!r}. Available: {names}")


def load_scene(path: str, *, device: Optional[Union[str, torch.device]] = None, dtype: torch.dtype = torch.float32,
               sample_rays: bool = True, nbounces: Optional[int] = None) -> Scene:
    """Instantiate every element and bundle of a project on `device` (default: the HIP device when there is
    one), register them on a new Scene and, unless told not to, sample the initial rays (project.py:96-179).
    The scene also gets ``element_names`` / ``bundle_names`` and ``find_element(name)`` / ``find_bundle(name)``."""
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu") if device is None else torch.device(device)
    data = load_project(path)
    scene = Scene()
    names = {"elements": [], "bundles": []}
    for kind in ("elements", "bundles"):
        for item in data.get(kind, []):
            cfg = item.get("config", item)
            obj = instantiate_from_config(cfg, device=device, dtype=dtype)
            if isinstance(obj, nn.Module):
                obj.to(device)
            if kind == "elements":
                scene.add_element(obj)
            else:
                scene.add_bundle(obj, int(item.get("N_rays", 200)))
            names[kind].append(cfg.get("name", ""))
    scene.Nbounces = int(nbounces if nbounces is not None else data.get("settings", {}).get("Nbounces", 100))
    scene.to(device)
    scene._build_index_maps()
    if sample_rays:
        scene._build_rays()
        if scene.rays is not None and hasattr(scene.rays, "to"):
            scene.rays = scene.rays.to(device)
    scene.element_names, scene.bundle_names = names["elements"], names["bundles"]
    scene.find_element = lambda name, _s=scene: _lookup(_s.elements, _s.element_names, name, "element")
    scene.find_bundle = lambda name, _s=scene: _lookup(_s.bundles, _s.bundle_names, name, "bundle")
    return scene
