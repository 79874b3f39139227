"""Host-side profile of Scene.simulate() (the fused bounce loop) and of a Scene.step() loop on the scene of the
reference's benchmark script (benchmarks/sim_benchmark.py:56-95) at a launch-bound size.  Run on the GPU box."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import raytracetorch_amd as rta  # noqa: E402
import scenes  # noqa: E402
import sim_benchmark as SB  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = sys.argv[2] if len(sys.argv) > 2 else "fused"
ns = scenes.namespace_from_package(rta)
scene, bundle = SB.build_scene(ns, torch.device("cuda", 0))
scene.add_bundle(bundle, n)
scene.Nbounces = 20
scene._build_rays()
start = scene.rays.clone()


def once():
    scene.rays = start.clone()
    if mode == "fused":
        scene.simulate()
    else:
        scene._build_index_maps()
        for _ in range(20):
            if not (scene.rays.intensity > 0).any():
                break
            scene.step()


with torch.no_grad():
    for _ in range(20):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        once()
    torch.cuda.synchronize()
    print("us per run:", (time.perf_counter() - t0) / 300 * 1e6)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        once()
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
