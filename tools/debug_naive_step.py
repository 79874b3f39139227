"""GPU: the worst rays of the full-size 32-surface step against the float32 and float64 oracle (which surface, how far
the oracle's own two precisions are apart there)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import raytracetorch_amd as rta
import raytracetorch_amd.program as P
import scenes
import rtt_oracle as O
DEV = torch.device("cuda")
import math


def device_disk_rays(ns, n, seed, radius=4.0, z=-30.0, tilt=0.0):
    g = torch.Generator(device=DEV); g.manual_seed(seed)
    theta = torch.rand(n, device=DEV, generator=g) * (2 * math.pi)
    r = torch.sqrt(torch.rand(n, device=DEV, generator=g) * (radius * radius))
    pos = torch.stack([r * torch.cos(theta), r * torch.sin(theta), torch.full_like(r, z)]).contiguous()
    d = torch.zeros((3, n), device=DEV)
    d[1].fill_(math.sin(tilt)); d[2].fill_(math.cos(tilt))
    return ns.Rays.from_planes(pos, d, torch.ones(n, device=DEV), torch.zeros(n, device=DEV, dtype=torch.int8),
                               torch.zeros(n, device=DEV))


ns = scenes.namespace_from_package(rta)
n = 4 * 2 ** 20
scene = ns.Scene()
for el in scenes.naive32(ns):
    scene.add_element(el)
prog = P.compile_elements(scene.elements)
rays = device_disk_rays(ns, n, seed=13, radius=6.5, z=-10.0, tilt=0.03)
h = n // 2
rays._pos[2, h:] = 44.5
gen = torch.Generator(device=DEV); gen.manual_seed(int(os.environ.get('SEED', '5')))
rays._dir[:, h:] = torch.nn.functional.normalize(torch.randn(3, n - h, device=DEV, generator=gen), dim=0)
before = rays.clone()
m = int(os.environ.get('SAMPLES', '131072'))
idx = (torch.arange(0, h, h // m, device=DEV)[:m] if os.environ.get('HALF') == 'base' else torch.arange(h, n, (n - h) // m, device=DEV)[:m])
p, d, I = before.pos[idx].cpu().numpy(), before.dir[idx].cpu().numpy(), before.intensity[idx].cpu().numpy()
scene.rays = rays
scene.step()
o = O.step_nearest(prog, p, d, I)
o64 = O.step_nearest(prog, p, d, I, dtype=torch.float64)
same = (o64["winner"].numpy() == o["winner"].numpy())
for f in ("pos", "dir"):
    got = getattr(rays, f)[idx].cpu().numpy()
    ref, tru = o[f].numpy(), o64[f].numpy()
    scale = np.maximum(np.linalg.norm(ref, axis=1), 1e-30)
    e_ref = np.linalg.norm(got - ref, axis=1) / scale
    e_tru = np.linalg.norm(got - tru, axis=1) / scale
    r_tru = np.linalg.norm(ref - tru, axis=1) / scale
    e_ref[~same] = 0
    worst = np.argsort(-e_ref)[:6]
    print(f, "rays over 1e-5:", int((e_ref > 1e-5).sum()))
    for k in worst:
        w = int(o["winner"].numpy()[k])
        kinds = (int(prog["surf_kind"][w]), int(prog["phys_kind"][w])) if w >= 0 else None
        print(f"  ray {k}: |got-ref32|={e_ref[k]:.3g} |got-ref64|={e_tru[k]:.3g} |ref32-ref64|={r_tru[k]:.3g} winner={w} kinds={kinds} "
              f"d_in={d[k]} out={got[k]} ref={ref[k]}")
