"""How coherent the wavefronts of the 32-surface scene stay over the bounces of Scene.simulate(): per bounce, the share of
rays still moving, and per wavefront (128 consecutive rays) the number of distinct winning surfaces and elements."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import raytracetorch_amd as rta
from raytracetorch_amd import kernels as K
import scenes
ns = scenes.namespace_from_package(rta)
dev = "cuda:0"
n = 1 << 20
scene = ns.Scene()
for el in scenes.naive32(ns):
    scene.add_element(el)
b = ns.CollimatedDisk(radius=4.0, ray_id=0, device=dev, transform=ns.RayTransformBundle(translation=[0.0, 0.0, -10.0]).to(dev))
rays = b.sample(n, seed=1000, offset=0)
rays._pos
prog = scene._program()
elem = torch.from_numpy(np.asarray(prog["element_id"]).astype(np.int64)).to(dev)
open_ = torch.ones(n, dtype=torch.bool, device=dev)
for bounce in range(int(os.environ.get("BOUNCES", 40))):
    act = open_ & (rays.intensity > 0)
    winner, *_ = K.step_nearest(rays, prog, want_winner=True)
    won = winner >= 0
    open_ &= ~(act & ~won)
    w = torch.where(act & won, winner.long(), torch.full_like(winner, -1).long()).view(-1, 128)
    e = torch.where(w >= 0, elem[w.clamp_min(0)], torch.full_like(w, -1))
    def distinct(x):
        s, _ = torch.sort(x, dim=1)
        new = (s[:, 1:] != s[:, :-1]) & (s[:, 1:] >= 0)
        return (new.sum(1) + (s[:, 0] >= 0)).float()
    ds, de = distinct(w), distinct(e)
    live_waves = (w >= 0).any(1)
    top = torch.bincount(winner[act & won].long(), minlength=32)
    print(f"bounce {bounce:3d}: moving {float((act & won).float().mean()):.3f}  waves with work {float(live_waves.float().mean()):.3f}  "
          f"distinct surfaces/wave {float(ds[live_waves].mean()) if live_waves.any() else 0:.2f}  elements/wave {float(de[live_waves].mean()) if live_waves.any() else 0:.2f}  "
          f"top surfaces {[(int(i), int(c)) for c, i in zip(*torch.topk(top, 4))]}", flush=True)
