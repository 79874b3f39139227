"""Cost of the tangent kernel versus the number of parameter directions (run on the GPU box)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import raytracetorch_amd as rta  # noqa: E402
import scenes  # noqa: E402

ns = scenes.namespace_from_package(rta)
dev = torch.device("cuda", 0)
n = 16 * 2 ** 20
bundle = ns.CollimatedDisk(radius=4.0, ray_id=0, device=dev, transform=ns.RayTransformBundle(translation=[0.0, 0.0, -30.0]).to(dev))
rays = bundle.sample(n, seed=1)


class Resident(torch.nn.Module):
    def sample(self, _n):
        return ns.Rays.from_planes(rays._pos, rays._dir, rays.intensity, rays.id, rays.wavelength)


for label, kw in (("c1 (K=2)", dict(c1_grad=True)), ("c1,c2 (K=2)", dict(c1_grad=True, c2_grad=True)),
                  ("c1,c2,t (K=4)", dict(c1_grad=True, c2_grad=True, t_grad=True)),
                  ("c1,c2,t,ior (K=6)", dict(c1_grad=True, c2_grad=True, t_grad=True, ior_glass_grad=True)),
                  ("c1,c2,t,ior,ior_media, pose xy (K=8)", dict(c1_grad=True, c2_grad=True, t_grad=True, ior_glass_grad=True,
                                                               ior_media_grad=True, pose=True))):
    if kw.pop("pose", False):
        kw["transform"] = ns.RayTransform(translation=[0.0, 0.0, 0.0], trans_grad=True, trans_mask=[True, True, False])
    lens = ns.SingletLens(c1=0.05, c2=-0.05, d=10.0, t=3.0, ior_glass=1.5, inked=True, **kw).to(dev)
    sensor = ns.Sensor(ns.Disk(6.0, transform=ns.RayTransform(translation=[0.0, 0.0, 19.0]))).to(dev)
    scene = ns.SequentialScene([lens, sensor])
    loss_fn = ns.SpotSizeLoss(sensor, [Resident()], N_rays=n)

    def step():
        for p in scene.parameters():
            p.grad = None
        loss_fn(scene).backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    print(f"{label:22s} {((time.perf_counter() - t0) / 10) * 1e3:7.3f} ms/step ({n} rays)")
