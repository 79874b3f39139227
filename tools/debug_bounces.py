import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
import raytracetorch_amd as rta, raytracetorch_amd.program as P, rtt_oracle as O, scenes
ns = scenes.namespace_from_package(rta)
scene = ns.Scene()
for el in (scenes.singlet(ns), ns.Sensor(ns.Disk(6.0, transform=scenes.T(ns, 19.0)))):
    scene.add_element(el)
torch.manual_seed(0)
scene.add_bundle(scenes.disk_bundle(ns), 4096)
scene._build_rays()
prog = P.compile_elements(scene.elements)
cpu = scene.rays.clone()
scene.rays = scene.rays.to("cuda:0")
p, d, I = cpu.pos.clone(), cpu.dir.clone(), cpu.intensity.clone()
for b in range(6):
    o = O.step_nearest(prog, p, d, I)
    scene.step()
    g = scene.rays
    gp, gd = g.pos.cpu(), g.dir.cpu()
    bad = ~torch.isfinite(gp).all(1)
    obad = ~torch.isfinite(o["pos"]).all(1)
    diff = (gp - o["pos"]).abs().max(1)[0]
    print(f"bounce {b}: gpu nonfinite {int(bad.sum())} oracle nonfinite {int(obad.sum())} max diff (finite) {float(diff[~bad & ~obad].max()):.3g} winners {np.unique(o['winner'].numpy(), return_counts=True)}")
    if bad.any():
        i = int(torch.nonzero(bad)[0])
        print("  ray", i, "in p", p[i].tolist(), "d", d[i].tolist(), "I", float(I[i]), "oracle win", int(o["winner"][i]), "tmin", float(o["tmin"][i]))
        print("  gpu out", gp[i].tolist(), gd[i].tolist(), " oracle out", o["pos"][i].tolist(), o["dir"][i].tolist())
    # continue from the GPU state so both see the same inputs
    p, d, I = gp.clone(), gd.clone(), g.intensity.cpu().clone()
