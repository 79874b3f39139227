"""Upper bound for regrouping rays by liveness in the 5-element system: the same bundle traced in its random order and
sorted by radius (so that the rays the stop kills sit in wavefronts of their own)."""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import raytracetorch_amd as rta
import scenes
ns = scenes.namespace_from_package(rta)
dev = torch.device("cuda", 0)
n = 16 * 2 ** 20
b = ns.CollimatedDisk(radius=4.0, ray_id=0, device=dev, transform=ns.RayTransformBundle(translation=[0.0, 0.0, -30.0]).to(dev))
r = b.sample(n, seed=1000)
order = torch.argsort(r._pos[0] ** 2 + r._pos[1] ** 2)
mk = lambda p, q: ns.Rays.from_planes(p, q, torch.ones(n, device=dev), torch.zeros(n, device=dev, dtype=torch.int8), torch.zeros(n, device=dev))
for label, rays in (("random order", mk(r._pos, r._dir)), ("sorted by radius", mk(r._pos[:, order].contiguous(), r._dir[:, order].contiguous()))):
    out = mk(torch.empty_like(r._pos), torch.empty_like(r._dir))
    els = scenes.system5(ns)
    scene = ns.SequentialScene(els)
    ts = []
    for i in range(25):
        els[-1].reset()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); scene.simulate(rays, out=out); e.record(); torch.cuda.synchronize()
        if i >= 5:
            ts.append(s.elapsed_time(e))
    print(f"{label:18s}: {statistics.median(ts)*1e3:7.1f} us  alive {float((out.intensity > 0).float().mean()):.4f}", flush=True)
