"""Pinhole / orbit cameras and the single-bounce shaded renderer (reference: render/camera.py:16-362).

``Renderer.render_3d`` is one kernel (``rtt_render_f32``): nearest hit over the scene's surface program with
the aperture elements left out, normal of the winning surface, base colour of its physics functor times a
two-sided Lambert term.  The cameras are host-side vector algebra on a handful of 3-vectors.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .. import _native as N
from .. import kernels as K
from .. import program as P
from ..rays.ray import Rays


class Camera:
    """Pinhole camera: one ray per pixel through an image plane at unit distance (render/camera.py:16-75)."""

    def __init__(self, position, look_at, up_vector, fov_deg, width, height, device="cpu"):
        self.device = device
        self.width, self.height, self.fov_deg = width, height, fov_deg
        self.origin = torch.tensor(position, dtype=torch.float32, device=device)
        target = torch.tensor(look_at, dtype=torch.float32, device=device)
        up = torch.tensor(up_vector, dtype=torch.float32, device=device)
        self.forward = F.normalize(target - self.origin, dim=0)
        self.right = F.normalize(torch.linalg.cross(self.forward, up), dim=0)
        self.up_cam = torch.linalg.cross(self.right, self.forward)

    def generate_rays(self) -> Rays:
        aspect = self.width / self.height
        scale_y = torch.tan(torch.deg2rad(torch.tensor(self.fov_deg * 0.5)))
        scale_x = scale_y * aspect
        y_grid = torch.linspace(scale_y, -scale_y, self.height, device=self.device)
        x_grid = torch.linspace(-scale_x, scale_x, self.width, device=self.device)
        yy, xx = torch.meshgrid(y_grid, x_grid, indexing="ij")
        dirs = xx.reshape(-1, 1) * self.right + yy.reshape(-1, 1) * self.up_cam + self.forward
        return Rays.initialize(self.origin.expand_as(dirs), dirs, device=self.device)


class OrbitCamera(Camera):
    """CAD-style camera orbiting a pivot with a level horizon (render/camera.py:78-168)."""

    def __init__(self, pivot=(0, 0, 0), **kwargs):
        super().__init__(**kwargs)
        self.pivot = torch.tensor(pivot, dtype=torch.float32, device=kwargs.get("device", "cpu"))
        self.update_view_matrix()

    def _vec(self, v):
        return torch.tensor(v, device=self.device)

    def update_view_matrix(self):
        direction = self.pivot - self.origin
        dist = torch.norm(direction)
        if dist < 1e-3:
            return
        self.forward = direction / dist
        self.right = torch.linalg.cross(self.forward, self._vec([0.0, 1.0, 0.0]))
        if torch.norm(self.right) < 1e-3:   # looking straight up or down
            self.right = self._vec([1.0, 0.0, 0.0])
        self.right = F.normalize(self.right, dim=0)
        self.up_cam = F.normalize(torch.linalg.cross(self.right, self.forward), dim=0)

    def orbit(self, d_yaw, d_pitch):
        def rotate(vec, axis, angle):   # Rodrigues
            angle = torch.as_tensor(angle, dtype=torch.float32, device=self.device)
            c, s = torch.cos(angle), torch.sin(angle)
            return vec * c + torch.linalg.cross(axis, vec) * s + axis * torch.dot(axis, vec) * (1 - c)

        world_up = self._vec([0.0, 1.0, 0.0])
        radius = rotate(self.origin - self.pivot, world_up, -d_yaw)
        unit = F.normalize(radius, dim=0)
        if torch.abs(torch.dot(unit, world_up)) > 0.95:
            axis = self._vec([1.0, 0.0, 0.0])
        else:
            axis = F.normalize(torch.linalg.cross(unit, world_up), dim=0)
        self.origin = self.pivot + rotate(radius, axis, d_pitch)
        self.update_view_matrix()

    def roll(self, angle):
        angle = torch.as_tensor(angle, dtype=torch.float32, device=self.device)
        c, s = torch.cos(angle), torch.sin(angle)
        self.right, self.up_cam = c * self.right - s * self.up_cam, s * self.right + c * self.up_cam

    def pan(self, dx, dy):
        move = self.right * -dx + self.up_cam * dy
        self.origin = self.origin + move
        self.pivot = self.pivot + move

    def zoom(self, delta):
        radius = self.origin - self.pivot
        scale = 1.0 - delta * 0.1
        if torch.norm(radius) * scale < 0.1:
            scale = 1.0
        self.origin = self.pivot + radius * scale


def _lerp(a, b, w):
    return tuple(x + (y - x) * w for x, y in zip(a, b))


def base_color(phys_func):
    """Colour of a surface by the class name of its functor (render/camera.py:266-301)."""
    name = type(phys_func).__name__
    if "Reflect" in name:
        return (1.0, 0.6, 0.0)
    if "Block" in name:
        return (0.2, 0.2, 0.2)
    if "Transmit" in name:
        return (0.0, 0.8, 0.2)
    if "RefractSnell" in name or "RefractFresnel" in name:
        n_val = max(float(P.scalar(getattr(phys_func, "ior_in", 1.5))), float(P.scalar(getattr(phys_func, "ior_out", 1.5))))
        white, cyan, blue = (0.9, 0.9, 0.9), (0.0, 1.0, 1.0), (0.3, 0.6, 1.0)
        navy, purple = (0.0, 0.0, 0.5), (0.3, 0.0, 0.3)
        if n_val <= 1.0:
            return white
        if n_val <= 1.3:
            return _lerp(white, cyan, (n_val - 1.0) / 0.3)
        if n_val <= 1.4:
            return _lerp(cyan, blue, (n_val - 1.3) / 0.1)
        if n_val <= 1.7:
            return _lerp(blue, navy, (n_val - 1.4) / 0.3)
        return _lerp(navy, purple, min((n_val - 1.7) / 0.3, 1.0))
    return (1.0, 0.0, 1.0)


def _is_aperture(el):
    return any("ApertureFilter" in type(f).__name__ or "Fuzzy" in type(f).__name__ for f in el.surface_functions)


class Renderer:
    """Single-bounce view of a scene with fixed directional shading (render/camera.py:171-318)."""

    def __init__(self, scene, background_color=(1.0, 1.0, 1.0), light_dir=(-0.5, 1.0, -1.0)):
        self.scene = scene
        self.bg_color = torch.tensor(background_color, dtype=torch.float32)
        self.light_dir = F.normalize(torch.as_tensor(light_dir, dtype=torch.float32), dim=0)

    def render_3d(self, camera) -> torch.Tensor:
        """[height, width, 3] image on the CPU, values in [0, 1]."""
        self.scene._build_index_maps()
        rays = camera.generate_rays()
        dev = rays.device
        n = len(rays)
        prog = self.scene._program()
        if len(prog) > 64:   # (the kernel takes the surfaces to leave out as a 64-bit mask)
            raise NotImplementedError(f"the renderer shades scenes of at most 64 surfaces (this one has {len(prog)})")
        base = np.ones((max(len(prog), 1), 3), dtype=np.float32)
        skip = 0
        for s, rec in enumerate(prog):
            el = self.scene.elements[int(rec["element_id"])]
            if _is_aperture(el):
                skip |= 1 << s
            else:
                base[s] = base_color(el.surface_functions[int(rec["surface_id"])])
        if len(prog) == 0 or skip == (1 << len(prog)) - 1:
            return self.bg_color.expand(n, 3).clone().reshape(camera.height, camera.width, 3)
        N.require_device(rays._pos)
        import ctypes
        rgb = torch.empty((n, 3), device=dev, dtype=torch.float32)
        base_dev = torch.from_numpy(base).to(dev)
        dp = N.DeviceProgram.upload(prog, dev)
        f3 = lambda t: (ctypes.c_float * 3)(*[float(v) for v in t.tolist()])
        with torch.cuda.device(dev):
            code = N.lib().rtt_render_f32(N.ptr(rays._pos), N.ptr(rays._dir), n, n, N.ptr(dp.dev), dp.host_ptr, dp.n,
                                          N.ptr(base_dev), ctypes.c_uint64(skip), f3(self.light_dir), f3(self.bg_color),
                                          N.ptr(rgb), N.stream_handle(dev))
        N.check(code, "rtt_render_f32")
        return rgb.reshape(camera.height, camera.width, 3).cpu()

    def scan_profile(self, target_element, axis="x", num_points=200, bounds=None):
        """Cross-section (height, z) samples of every surface of one element (render/camera.py:320-362)."""
        device = target_element.shape.transform.trans.device
        if bounds:
            extent, center = bounds[1] - bounds[0], (bounds[0] + bounds[1]) / 2
        else:
            radius = getattr(target_element.shape, "radius", torch.tensor(10.0))
            extent, center = float(P.scalar(radius)) * 2.2, 0.0
        coords = torch.linspace(center - extent / 2, center + extent / 2, num_points, device=device)
        zeros, z_start = torch.zeros_like(coords), torch.full_like(coords, -100.0)
        origins = torch.stack([coords, zeros, z_start] if axis == "x" else [zeros, coords, z_start], dim=1)
        directions = torch.tensor([0.0, 0.0, 1.0], device=device).expand(num_points, 3)
        rays = Rays.initialize(origins, directions, device=device)
        t_matrix = target_element.intersectTest(rays)
        results = []
        for i in range(t_matrix.shape[1]):
            t = t_matrix[:, i]
            mask = t < float("inf")
            if mask.any():
                results.append({"surf_idx": i, "h": coords[mask].cpu().numpy(), "z": (-100.0 + t[mask]).cpu().numpy()})
        return results
