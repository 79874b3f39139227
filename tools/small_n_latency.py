"""Per-call cost of SequentialScene.simulate and Scene.step on small bundles (host overhead + launch)."""
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
dev = "cuda:0"
for name, els in (("plane", [ns.ElementCustom(ns.Plane(transform=ns.RayTransform()), ns.Transmit(), dev)]),
                  ("singlet+sensor", list(scenes.spot_system(ns, grads=False))),
                  ("system5", scenes.system5(ns))):
    for n in (1024, 2 ** 20):
        rays = scenes.disk_bundle(ns).sample(n).to(dev)
        scene = ns.SequentialScene(els)
        out = rays.clone()
        with torch.no_grad():
            for _ in range(20):
                scene.simulate(rays, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 500
            for _ in range(reps):
                for el in els:   # a sensor keeps every visit until it is reset, like the reference's
                    if hasattr(el, "reset"):
                        el.reset()
                scene.simulate(rays, out=out)
            torch.cuda.synchronize()
        print(f"{name:16s} n={n:8d}  {1e6 * (time.perf_counter() - t0) / reps:8.1f} us per simulate", flush=True)
