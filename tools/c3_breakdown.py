"""Marginal kernel time of each element of the 5-element system: the sequential trace of its first k elements,
k = 1..5, over the bench bundle (16 Mi rays), HIP-event time per launch (median of 20)."""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import raytracetorch_amd as rta
import scenes
ns = scenes.namespace_from_package(rta)
dev = torch.device("cuda", 0)
n = int(os.environ.get("RAYS", 16 * 2 ** 20))
b = ns.CollimatedDisk(radius=4.0, ray_id=0, device=dev, transform=ns.RayTransformBundle(translation=[0.0, 0.0, -30.0]).to(dev))
r = b.sample(n, seed=1000)
mk = lambda p, q: ns.Rays.from_planes(p, q, torch.ones(n, device=dev), torch.zeros(n, device=dev, dtype=torch.int8), torch.zeros(n, device=dev))
rays = mk(r._pos, r._dir)
out = mk(torch.empty_like(r._pos), torch.empty_like(r._dir))
prev = 0.0
for k in range(1, 6):
    els = scenes.system5(ns)[:k]
    scene = ns.SequentialScene(els)
    ts = []
    for i in range(25):
        if k == 5:
            els[-1].reset()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); scene.simulate(rays, out=out); e.record(); torch.cuda.synchronize()
        if i >= 5:
            ts.append(s.elapsed_time(e))
    t = statistics.median(ts)
    alive = int((out.intensity > 0).sum())
    print(f"first {k} elements: {t*1e3:8.1f} us  (+{(t-prev)*1e3:7.1f})  alive {alive/n:.4f}", flush=True)
    prev = t
