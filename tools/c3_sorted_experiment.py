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
rad2 = r._pos[0] ** 2 + r._pos[1] ** 2
order = torch.argsort(rad2)
# what a regrouping inside the kernel could reach at best: the rays of every tile of `tile` consecutive rays (512 = one
# block, 128 = one wavefront of two rays per lane) ordered by radius, i.e. live rays first within the tile, at no cost
def within_tiles(tile):
    key = torch.arange(n, device=dev) // tile * 1024.0 + rad2.double()      # (radius^2 <= 16)
    return torch.argsort(key)
mk = lambda p, q: ns.Rays.from_planes(p, q, torch.ones(n, device=dev), torch.zeros(n, device=dev, dtype=torch.int8), torch.zeros(n, device=dev))
pick = lambda o: mk(r._pos[:, o].contiguous(), r._dir[:, o].contiguous())
dead = rad2 > 3.5 ** 2                                                      # (what the inverted stop of radius 3.5 kills, about)
for tile in (128, 512):
    d = dead[within_tiles(tile)].view(-1, 128).float().mean(dim=1)
    print(f"tiles of {tile}: wavefronts all live {float((d == 0).float().mean()):.3f}, all dead {float((d == 1).float().mean()):.3f}, "
          f"mixed {float(((d > 0) & (d < 1)).float().mean()):.3f}", flush=True)
for label, rays in (("random order", mk(r._pos, r._dir)), ("sorted in tiles of 512", pick(within_tiles(512))),
                    ("sorted in tiles of 4096", pick(within_tiles(4096))), ("sorted by radius", pick(order))):
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
