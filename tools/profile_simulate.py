"""Host-side profile of SequentialScene.simulate for a launch-bound bundle (1024 rays).  Run on the GPU box."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import raytracetorch_amd as rta  # noqa: E402
import scenes  # noqa: E402

ns = scenes.namespace_from_package(rta)
scene = ns.SequentialScene([scenes.singlet(ns)])
torch.manual_seed(0)
rays = scenes.disk_bundle(ns).sample(1024).to("cuda:0")
out = rays.clone()
for _ in range(100):
    scene.simulate(rays, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5000):
    scene.simulate(rays, out=out)
torch.cuda.synchronize()
print("us per simulate:", (time.perf_counter() - t0) / 5000 * 1e6)
pr = cProfile.Profile()
pr.enable()
for _ in range(5000):
    scene.simulate(rays, out=out)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
