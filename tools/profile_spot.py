"""Host-side profile of the spot-size optimisation step (bench.py --workload spot): where the time between
kernels goes.  Run on the GPU box: python tools/profile_spot.py"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import raytracetorch_amd as rta  # noqa: E402
import scenes  # noqa: E402

ns = scenes.namespace_from_package(rta)
dev = torch.device("cuda", 0)
n = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 16 * 2 ** 20
lens, sensor = scenes.spot_system(ns, grads=True)
scene = ns.SequentialScene([lens, sensor])
if "--device-params" in sys.argv:   # parameters resident on the GPU, as load_scene leaves them
    scene.to(dev)
bundle = ns.CollimatedDisk(radius=4.0, ray_id=0, device=dev, transform=ns.RayTransformBundle(translation=[0.0, 0.0, -30.0]).to(dev))
rays = bundle.sample(n, seed=1)


class Resident(torch.nn.Module):
    def sample(self, _n):
        return ns.Rays.from_planes(rays._pos, rays._dir, rays.intensity, rays.id, rays.wavelength)


loss_fn = ns.SpotSizeLoss(sensor, [Resident()], N_rays=n)


opt = torch.optim.SGD([p for p in scene.parameters() if p.requires_grad], lr=0.0) if "--optimizer" in sys.argv else None


def step():
    for p in scene.parameters():
        p.grad = None
    loss_fn(scene).backward()
    if opt is not None:   # a real loop changes the parameters every step: the program is re-packed each time
        opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 20 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
pstats.Stats(pr).sort_stats("cumtime").print_stats("raytracetorch_amd|run_backward|tolist", 40)
